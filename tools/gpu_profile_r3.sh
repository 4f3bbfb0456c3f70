#!/usr/bin/env bash
# Round-3 profile set (gpurun -- 'bash tools/gpu_profile_r3.sh'); results under gpurun_out/prof_r3, copied into profiles/r3/ by hand.
set -u
REPO=$(pwd); O=$REPO/gpurun_out/prof_r3; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
python $REPO/bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- python $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-train-leg > $O/bench_under_rocprof.json 2> $O/rocprof.err
python $REPO/tools/kstats.py $(find $O/trace -name "*kernel_stats.csv" | head -1) 40 > $O/bench_kernel_stats_short.txt
rm -rf $O/trace
python $REPO/bench.py --train 3 --steps 10 --warmup 6 > $O/bench_train_cfg3.json 2>/dev/null
python $REPO/bench.py --train 4 --steps 10 --warmup 6 > $O/bench_train_cfg4.json 2>/dev/null
python $REPO/bench.py --train 3 --steps 10 --warmup 6 --precision train_f16x3 > $O/bench_train_cfg3_f16x3.json 2>/dev/null
python $REPO/bench.py --train 3 --steps 10 --warmup 6 --precision train_amp_bf16 --no-cpu-baseline > $O/bench_train_cfg3_amp_bf16.json 2>/dev/null
python $REPO/bench.py --train 3 --steps 10 --warmup 6 --precision train_amp_fp16 --no-cpu-baseline > $O/bench_train_cfg3_amp_fp16.json 2>/dev/null
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_train -o train -- python $REPO/bench.py --train 3 --steps 4 --warmup 4 --no-cpu-baseline --precision mixed > $O/train3_under_rocprof.json 2> $O/rocprof_train.err
python $REPO/tools/kstats.py $(find $O/trace_train -name "*kernel_stats.csv" | head -1) 70 > $O/train_cfg3_kernel_stats.txt
rm -rf $O/trace_train
python $REPO/tools/bench_wgrad.py --cfg 3 > $O/bench_wgrad_cfg3.txt 2>/dev/null
# PMC passes of the dominant backward kernel, alone (separate runs per counter set)
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -o k -- python $REPO/tools/run_wgrad_pk.py 4 > /dev/null 2> $O/pmc_$c.err
  f=$(find $O/pmc_$c -name "*counter_collection.csv" | head -1)
  grep -E "k_gemm_pk|Kernel_Name" "$f" > $O/pmc_wgrad_$c.csv
  rm -rf $O/pmc_$c
done
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU --output-format csv -d $O/pmc_sq -o k -- python $REPO/tools/run_wgrad_pk.py 4 > /dev/null 2> $O/pmc_sq.err
f=$(find $O/pmc_sq -name "*counter_collection.csv" | head -1)
[ -n "$f" ] && python $REPO/tools/kstats.py "$f" | grep -A12 "k_gemm_pk" > $O/pmc_wgrad_sq.txt
rm -rf $O/pmc_sq
python - <<PY
import csv, json, re
def mean(path):
    rows = [r for r in csv.DictReader(open(path)) if "k_gemm_pk" in r["Kernel_Name"]]
    v = [float(r["Counter_Value"]) for r in rows]
    name = re.sub(r"\(.*", "", rows[0]["Kernel_Name"]).replace("void craft::", "")
    return sum(v) / len(v), len(v), name
f, n, name = mean("$O/pmc_wgrad_FETCH_SIZE.csv")
w, _, _ = mean("$O/pmc_wgrad_WRITE_SIZE.csv")
json.dump({"kernel": "k_gemm_pk", "kernel_instantiation": name, "policy": "train_f16x3", "shape": [128, 256, 3, 3, 22816, 12],
           "shape_legend": "cin, cout, KH, KW, pixels per call, calls per launch",
           "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes over tools/run_wgrad_pk.py (tools/gpu_profile_r3.sh)",
           "fetch_kb_raw": f, "write_kb_raw": w, "fetch_correction": 2.0,
           "hbm_bytes_per_launch": int(f * 1024 * 2 + w * 1024), "launches_averaged": n}, open("$O/pmc_traffic_wgrad.json", "w"), indent=1)
PY
cat $O/pmc_traffic_wgrad.json; cat $O/pmc_wgrad_sq.txt 2>/dev/null | head -14
du -sh $O
# attention products / softmax kernels at configs[3]'s shapes, per-call and aten views of the step
python $REPO/tools/bench_gemm_pkb.py f16x3 > $O/bench_gemm_pkb_f16x3.txt 2>/dev/null
python $REPO/tools/bench_gemm_pkb.py bf16 > $O/bench_gemm_pkb_bf16.txt 2>/dev/null
python $REPO/tools/bench_softmax.py > $O/bench_softmax.txt 2>/dev/null
python $REPO/tools/call_profile.py 3 60 > $O/call_profile_cfg3.txt 2>/dev/null
python $REPO/tools/call_profile.py 4 60 > $O/call_profile_cfg4.txt 2>/dev/null
python $REPO/tools/aten_profile.py 3 > $O/aten_profile_cfg3.txt 2>/dev/null
cd $REPO && bash tools/gpu_pmc_pkb.sh f16x3 > /dev/null 2>&1
cp $REPO/gpurun_out/pmc_pkb/kernel_stats.txt $O/pmc_pkb_kernel_stats.txt; cp $REPO/gpurun_out/pmc_pkb/pmc_sq.txt $O/pmc_pkb_sq.txt; cp $REPO/gpurun_out/pmc_pkb/pmc_b.txt $O/pmc_pkb_grbm_lds.txt
du -sh $O
