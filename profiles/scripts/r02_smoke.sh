#!/bin/bash
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
timeout 1700 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -3
