# Does the dominant kernel run FASTER at a lower core clock?  (profiles/r04_power_under_load.txt: on a box that throttles to ~2040 MHz it takes
# 446 us on random operands, and 482 us at the full 2393 MHz on all-zero operands -- same instruction stream, same addresses.)
# Caps sclk with rocm-smi's performance-determinism mode and times the kernel (random / zero operands), the GEMM-free twins and the copy.
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-sclk}; mkdir -p $O
export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O3 scripts/adamprobe.hip -o /tmp/adamprobe 2>/dev/null
sed -n '/^cat > \/tmp\/loop.py/,/^PY$/p' scripts/clocks_under_load.sh | sed '1d;$d' > /tmp/loop.py
snap() { echo "$1 $(rocm-smi --showpower --showclocks 2>/dev/null | grep -i 'Package Power\|sclk\|fclk\|mclk' | sed 's/.*: //' | tr '\n' ' ')" >> $O/samples.txt; }
run() {  # $1 tag, rest = command
  tag=$1; shift
  "$@" > $O/out_$tag.txt 2>&1 &
  pid=$!
  sleep 3; snap $tag; sleep 1; snap $tag
  wait $pid
  grep -h "us per launch" $O/out_$tag.txt | sed "s/^/$tag /" >> $O/rates.txt
}
for cap in default 2200 2000 1800 1600 1400; do
  if [ $cap != default ]; then
    rocm-smi --setperfdeterminism $cap > $O/set_$cap.txt 2>&1 || echo "setperfdeterminism $cap failed" >> $O/rates.txt
  fi
  run cap${cap}_kernel python /tmp/loop.py kernel 5
  run cap${cap}_zero python /tmp/loop.py kernel_zero_operands 5
  run cap${cap}_pers /tmp/adamprobe r loop pers 5
  run cap${cap}_cols /tmp/adamprobe r loop cols 5
  run cap${cap}_copy python /tmp/loop.py copy 4
done
rocm-smi --resetperfdeterminism > $O/reset.txt 2>&1
run capreset_kernel python /tmp/loop.py kernel 5
cat $O/rates.txt
