cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
hipcc --offload-arch=gfx950 -O3 scripts/adamprobe.hip -o /tmp/adamprobe 2>/dev/null
/tmp/adamprobe r arena > $O/arena_1.txt 2>&1
/tmp/adamprobe r arena > $O/arena_2.txt 2>&1
/tmp/adamprobe r placement > $O/placement.txt 2>&1
cat $O/arena_1.txt; echo; cat $O/arena_2.txt | head -8; echo; head -8 $O/placement.txt
