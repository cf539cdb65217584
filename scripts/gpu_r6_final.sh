# Round-6 closing session: GPU tests, the driver's bench command, kernel stats, counters.  usage: bash scripts/gpu_r6_final.sh [notests] [nopmc]
cd $GRAFT_REPO_ROOT
bash scripts/gpu_round6.sh $1
if [ "$2" != nopmc ]; then bash scripts/gpu_pmc_round6.sh 2>&1 | tail -25; fi
