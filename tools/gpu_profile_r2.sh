#!/usr/bin/env bash
# Round-2 profile set (run on the GPU box: gpurun -- 'bash tools/gpu_profile_r2.sh'); results under gpurun_out/prof_r2, the
# summaries are copied into profiles/r2/ by hand.
#  1. rocprofv3 --kernel-trace --stats of the headline bench command            -> bench_kernel_stats_short.txt
#  2. the bench line itself (un-profiled)                                        -> bench.json
#  3. PMC passes (separate runs, FETCH_SIZE / WRITE_SIZE) of the dominant kernel -> pmc_traffic_pv16.json
#  4. kernel stats of the steady-state training step (configs[3])                 -> train_cfg3_kernel_stats.txt
#  5. per-kernel L2<->fabric traffic of the whole forward (FETCH_SIZE / WRITE_SIZE) -> traffic_table.txt
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/prof_r2; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
python $REPO/bench.py --steps 20 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/rocprof.err
python $REPO/tools/kstats.py $(find $OUT/trace -name "*kernel_stats.csv" | head -1) 40 > $OUT/bench_kernel_stats_short.txt
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_pv_$c -o k -- python $REPO/tools/run_kernel.py pv mixed > /dev/null 2> $OUT/pmc_pv_$c.err
  f=$(find $OUT/pmc_pv_$c -name "*counter_collection.csv" | head -1)
  grep -E "k_pv16|Kernel_Name" "$f" > $OUT/pmc_pv_$c.csv
done
python - <<PY
import csv, json, re
def mean(path):
    rows = list(csv.DictReader(open(path)))
    v = [float(r["Counter_Value"]) for r in rows if "k_pv16" in r["Kernel_Name"]]
    name = re.sub(r"\(.*", "", [r["Kernel_Name"] for r in rows if "k_pv16" in r["Kernel_Name"]][0]).replace("void craft::", "")
    return sum(v) / len(v), len(v), name
f, n, name = mean("$OUT/pmc_pv_FETCH_SIZE.csv")
w, _, _ = mean("$OUT/pmc_pv_WRITE_SIZE.csv")
json.dump({"kernel": name, "shape": {"B": 4, "H8": 56, "W8": 128, "M": 4, "Dv": 128, "p_dtype": "fp16"},
           "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes over tools/run_kernel.py pv (tools/gpu_profile_r2.sh)",
           "fetch_kb_raw": f, "write_kb_raw": w, "fetch_correction": 2.0,
           "hbm_bytes_per_launch": int(f * 1024 * 2 + w * 1024), "launches_averaged": n}, open("$OUT/pmc_traffic_pv16.json", "w"), indent=1)
PY
# steady-state training step: drop the first (MIOpen find) iteration by tracing only kernels of the last steps is not possible
# with --stats, so the table lists averages over all calls; MIOpen's one-off search kernels (naive_conv_*) are marked by name
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_train -o train -- python $REPO/bench.py --train 3 --steps 3 --warmup 1 > $OUT/train3_under_rocprof.json 2> $OUT/rocprof_train.err
python $REPO/tools/kstats.py $(find $OUT/trace_train -name "*kernel_stats.csv" | head -1) 60 > $OUT/train_cfg3_kernel_stats.txt
python $REPO/bench.py --train 3 --steps 5 --warmup 2 > $OUT/bench_train_cfg3.json 2>/dev/null
python $REPO/bench.py --train 4 --steps 5 --warmup 2 > $OUT/bench_train_cfg4.json 2>/dev/null
python $REPO/tools/bench_corr.py > $OUT/bench_corr_768x1024.json 2>/dev/null
CMD="python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tt0 -o k -- $CMD > /dev/null 2>&1
for c in FETCH_SIZE WRITE_SIZE; do timeout 400 rocprofv3 --pmc $c --output-format csv -d $OUT/tt_$c -o k -- $CMD > /dev/null 2>&1; done
python $REPO/tools/traffic_table.py $(find $OUT/tt0 -name "*kernel_stats.csv" | head -1) $(find $OUT/tt_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find $OUT/tt_WRITE_SIZE -name "*counter_collection.csv" | head -1) 32 > $OUT/traffic_table.txt
rm -rf $OUT/tt0 $OUT/tt_FETCH_SIZE $OUT/tt_WRITE_SIZE
rm -rf $OUT/trace $OUT/trace_train $OUT/pmc_pv_FETCH_SIZE $OUT/pmc_pv_WRITE_SIZE
du -sh $OUT
