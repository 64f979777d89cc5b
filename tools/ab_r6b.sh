# one-off visit: same-box A/B of the round-6b switches (TE_S2S6_FORM=0 TE_T2_EDGE=0 = before) in the training iteration + FFHQ-1024 sub-benchmark
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for i in 1 2; do
for f in old new; do
  if [ $f = old ]; then export TE_S2S6_FORM=0 TE_T2_EDGE=0; else unset TE_S2S6_FORM TE_T2_EDGE; fi
  timeout 300 python bench.py --steps 16 --warmup 4 --no-sub --no-cpu-baseline --no-pmc > gpurun_out/r6b_quick_$f.$i.json 2>/dev/null
done; done
for f in old new; do
  if [ $f = old ]; then export TE_S2S6_FORM=0 TE_T2_EDGE=0; else unset TE_S2S6_FORM TE_T2_EDGE; fi
  timeout 300 python bench.py --workload generator --size 1024 --steps 10 --warmup 4 --no-cpu-baseline --no-kernel-timing --no-pmc > gpurun_out/r6b_g1024_$f.json 2>/dev/null
  timeout 300 python bench.py --workload generator --steps 10 --warmup 4 --no-cpu-baseline --no-kernel-timing --no-pmc > gpurun_out/r6b_gen_$f.json 2>/dev/null
done
unset TE_S2S6_FORM TE_T2_EDGE
python - <<PY
import json
for i in (1,2):
  for f in ("old","new"):
    d=json.loads(open("gpurun_out/r6b_quick_%s.%d.json" % (f,i)).read().strip().splitlines()[-1]); print("quick A/B", f, d["value"], d["ms_per_step"], d["roofline"]["frac"])
for w in ("g1024","gen"):
  for f in ("old","new"):
    d=json.loads(open("gpurun_out/r6b_%s_%s.json" % (w,f)).read().strip().splitlines()[-1]); print(w, f, d["value"], d["ms_per_step"])
PY
