#!/bin/bash
timeout 1700 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -15
