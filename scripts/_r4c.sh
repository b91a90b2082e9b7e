cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_vae_chain.py -x -q -m gpu > $O/pytest_new.txt 2>&1
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_production.py tests/test_gpu_api.py tests/test_gpu_tail.py -x -q -m gpu -k "vae or svae or cross or cfg3 or supervised or golden or fullsize or random or fusion" > $O/pytest.txt 2>&1
Q="--no-cpu-baseline --sweep-trials-per-gpu 0 --no-other --repeats 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_cfg3 -- python bench.py --config cfg3 --steps 40 --warmup 5 $Q > $O/bench_prof_cfg3.json 2> $O/bench_prof_cfg3.err
cp $(find $O/trace_cfg3 -name "*kernel_trace.csv" | head -1) $O/kernel_trace_cfg3.csv
rm -rf $O/trace_cfg3
Q="--steps 30 --warmup 10 --no-cpu-baseline --sweep-trials-per-gpu 0 --no-other --repeats 5 --config cfg3"
for rep in 1 2 3; do
python bench.py $Q 2>/dev/null | tail -1 > $O/cfg3_new_$rep.json
FX_RECON_EPILOGUE=0 FX_VAE_LATENT_FUSED=0 FX_VAE_HEADS_BRANCH=0 FX_VAE_DEFER_MMD=0 FX_VAE_FUSION_PAIR=0 python bench.py $Q 2>/dev/null | tail -1 > $O/cfg3_old_$rep.json
done
FX_VAE_DEFER_MMD=0 python bench.py $Q 2>/dev/null | tail -1 > $O/cfg3_nodefer.json
FX_VAE_FUSION_PAIR=0 python bench.py $Q 2>/dev/null | tail -1 > $O/cfg3_nopair.json
