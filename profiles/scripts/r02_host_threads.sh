#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for t in 8 16 32 64; do echo "== CRYO_HOST_THREADS=$t"; CRYO_HOST_THREADS=$t timeout 600 python3 profiles/host_api_rate.py 2>&1 | grep "4096 x"; done
