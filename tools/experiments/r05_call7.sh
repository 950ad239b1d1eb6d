#!/bin/bash
# round 5, last GPU call: the round's records from HEAD (tools/final_round.sh), then a long soak with what is left
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
bash tools/final_round.sh r05_final2
timeout -k 5 780 python tools/soak2.py 120 9 2,3,6 > gpurun_out/r05_final2/soak2_long.txt 2>&1
tail -3 gpurun_out/r05_final2/soak2_long.txt; grep -c '^ok' gpurun_out/r05_final2/soak2_long.txt
