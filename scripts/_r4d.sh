cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest_all.txt 2>&1; echo "all rc=$? $(tail -1 $O/pytest_all.txt | cut -c1-80)" >> $O/summary.txt
K="vae or svae or cross or cfg3 or supervised or golden or fullsize or random or fusion or fit"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py -x -q -m gpu -p no:cacheprovider -k "$K" > $O/seq_s5.txt 2>&1; echo "s5 rc=$? $(tail -1 $O/seq_s5.txt | cut -c1-80)" >> $O/summary.txt
K="vae or svae or cross or cfg3 or supervised or golden or fullsize or random or fusion or reproducible or fit"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_production.py tests/test_gpu_api.py tests/test_gpu_vae_chain.py -x -q -m gpu -p no:cacheprovider -k "$K" > $O/seq_long.txt 2>&1; echo "long rc=$? $(tail -1 $O/seq_long.txt | cut -c1-80)" >> $O/summary.txt
for r in 1 2 3 4 5 6; do timeout 300 python scripts/race_vae.py graph 4 2>&1 | grep -v amdgpu >> $O/race.txt; done
cat $O/summary.txt; sort $O/race.txt | uniq -c
