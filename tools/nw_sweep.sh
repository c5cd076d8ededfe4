#!/bin/bash
for nw in 2 4 6 8 12 16; do echo "== NW $nw"; timeout -k 10 200 python tools/time3d.py 8 4 30 10 2 waves=$nw 2>&1 | grep -E "pass|k3_pg"; done
