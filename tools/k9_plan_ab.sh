line() { python3 -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-26s %8.2f Mpix/s  %.4f ms/step  attention %.4f ms  hbm %.4f' % (sys.argv[1], d['value'], d['ms_per_step'], r['kernel_ms'], r['frac']))" "$1"; }
for rep in 1 2; do
for w in G1-k9 G3-k9; do
  python bench.py --workload $w --steps 200 --no-cpu-baseline --no-live-traffic --no-cold-reading 2>/dev/null | tail -1 | line "$w default"
  for cap in 128 96 64; do
    NAF_HIP_KNOBS=1 NAF_XNA_DVT=$cap python bench.py --workload $w --steps 200 --no-cpu-baseline --no-live-traffic --no-cold-reading 2>/dev/null | tail -1 | line "$w DVT<=$cap"
  done
  NAF_HIP_KNOBS=1 NAF_XNA_STAGE=0 python bench.py --workload $w --steps 200 --no-cpu-baseline --no-live-traffic --no-cold-reading 2>/dev/null | tail -1 | line "$w unstaged (sliding)"
done
done
