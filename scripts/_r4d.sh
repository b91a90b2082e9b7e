cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
run() { tag=$1; shift; for r in $(seq 1 14); do env "$@" timeout 300 python scripts/race_vae.py graph 4 > $O/r_${tag}_$r.txt 2>&1; echo "$tag $r rc=$? diff=$(grep -c 'equal = False' $O/r_${tag}_$r.txt)" >> $O/summary.txt; done; }
run default A=1
run noheads FX_VAE_HEADS_BRANCH=0
cat $O/summary.txt
