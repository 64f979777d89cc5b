#!/bin/bash
# Evidence run (rounds 5 - 6) on the GPU box: full GPU test tier, smoke, default bench (the driver's command), RCCL world-1 rehearsal,
# kernel traces of the benchmarked workloads, PMC passes.   Tag of the outputs: $TAG (default r6).
TAG=${TAG:-r6}
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
( time timeout 1800 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider -rA ) > gpurun_out/${TAG}_pytest_gpu.log 2>&1
echo "pytest rc=$?"; grep -E "passed|failed" gpurun_out/${TAG}_pytest_gpu.log | tail -2; grep -E "^FAILED|^ERROR" gpurun_out/${TAG}_pytest_gpu.log | head
( timeout 300 python __graft_entry__.py --smoke ) > gpurun_out/${TAG}_smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/${TAG}_smoke.log
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/${TAG}_bench_n1.json 2> gpurun_out/${TAG}_bench_n1.err; echo "bench rc=$?"; cp gpurun_out/bench_detail.json gpurun_out/${TAG}_bench_detail.json
( TE_BENCH_FORCE_DIST=1 HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 300 python bench.py --gpus 1 --steps 4 --warmup 2 --no-sub --no-cpu-baseline --no-pmc ) > gpurun_out/${TAG}_bench_rccl_world1.json 2> gpurun_out/${TAG}_bench_rccl_world1.err; echo "bench RCCL world-1 rehearsal rc=$? lines=$(wc -l < gpurun_out/${TAG}_bench_rccl_world1.json)"
cd /tmp
( timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_train -o train -- python $R/bench.py --steps 16 --warmup 2 --no-cpu-baseline --no-sub --no-kernel-timing --no-pmc ) > $R/gpurun_out/${TAG}_rocprof_train.log 2>&1; echo "rocprof train rc=$?"
( timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_gen -o gen -- python $R/bench.py --workload generator --steps 8 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-pmc ) > $R/gpurun_out/${TAG}_rocprof_gen.log 2>&1; echo "rocprof gen rc=$?"
( timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_g1024 -o g1024 -- python $R/bench.py --workload generator --size 1024 --steps 6 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-pmc ) > $R/gpurun_out/${TAG}_rocprof_g1024.log 2>&1; echo "rocprof 1024 rc=$?"
cd $R
for t in train gen g1024; do python tools/rocpd_stats.py gpurun_out/prof_$t/${t}_results.db > gpurun_out/${TAG}_${t}_kernel_stats.txt 2>&1; rm -rf gpurun_out/prof_$t; done
rm -rf gpurun_out/pmc
bash tools/pmc_round.sh > gpurun_out/${TAG}_pmc_round.log 2>&1
python tools/pmc_summary.py gpurun_out/pmc gpurun_out/${TAG} > gpurun_out/${TAG}_pmc_summary_stdout.txt 2>&1
rm -f gpurun_out/pmc/*.db gpurun_out/pmc/*.csv
( FORM2=1 timeout 120 python tools/wino6_ab.py; timeout 120 python tools/s2s6_check.py; timeout 120 python tools/t2s6_check.py; timeout 200 python tools/wgrad6_check.py; timeout 200 python tools/p1s6_check.py ) 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_split_kernels_check.log
for f in old new; do
  if [ $f = old ]; then export TE_W6_FORM=1 TE_SPLIT_1X1=0 TE_S2S6_FORM=0 TE_T2S6_FORM=0 TE_T2_EDGE=0 TE_WGRAD_T2_WIDE=0; else unset TE_W6_FORM TE_SPLIT_1X1 TE_S2S6_FORM TE_T2S6_FORM TE_T2_EDGE TE_WGRAD_T2_WIDE; fi
  timeout 300 python bench.py --steps 16 --warmup 4 --no-sub --no-cpu-baseline --no-pmc > gpurun_out/${TAG}_bench_quick_r6_switches_$f.json 2>/dev/null
done
unset TE_W6_FORM TE_SPLIT_1X1 TE_S2S6_FORM TE_T2S6_FORM TE_T2_EDGE TE_WGRAD_T2_WIDE
( timeout 120 python tools/power_probe.py ) > gpurun_out/${TAG}_power_clock_product.txt 2>&1
python - <<PY
import json
for f in ("old","new"):
    d=json.loads(open("gpurun_out/${TAG}_bench_quick_r6_switches_%s.json" % f).read().strip().splitlines()[-1]); print("quick A/B", f, d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"])
PY
python - <<PY
import json
t=open("gpurun_out/${TAG}_bench_n1.json").read().strip().splitlines()
d=json.loads(t[-1])
r=d["roofline"]
print("bench line:", len(t), "line(s),", len(t[-1]), "bytes")
print("bench", d["value"], d["ms_per_step"], "dominant", r["kernel"], "frac", r["frac"], "alg", r["achieved_algorithmic"], "all launches", r.get("frac_all_launches"), r.get("mfma_util_pct"), r.get("mhz"), r.get("traffic_source"), r.get("traffic_over_algorithmic"))
print(d["substeps"])
for k,v in d.get("sub_benchmarks",{}).items(): print(k, v)
print(d.get("cpu_baseline"))
PY
grep TOTAL gpurun_out/${TAG}_train_kernel_stats.txt gpurun_out/${TAG}_gen_kernel_stats.txt gpurun_out/${TAG}_g1024_kernel_stats.txt
