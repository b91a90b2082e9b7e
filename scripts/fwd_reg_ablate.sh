# timing ablations of fx_fwd_bf16x3_reg_kernel (build the variants first: see profiles/r05_fwd_reg.txt)
for n in ${FX_ABL:-full noa now nostage nordb nobar nomem noall}; do echo "== $n"; FXHIP_LIB=build_tmp/libfxhip_$n.so FX_AB_ROWS=2 timeout 120 python scripts/fwd_reg_ab.py 2>&1 | grep "M=" ; done
