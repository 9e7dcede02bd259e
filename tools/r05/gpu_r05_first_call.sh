#!/bin/bash
# round 5, GPU call 1: the new tests, the bench line with the ViT-L leg, the clock PMC pass, FR = 32 pipelined A/B
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r05a; mkdir -p $O; cd $R
timeout 900 python -m pytest tests -m gpu -q -s -p no:cacheprovider -k "769_views or rejects_lab or beyond_2p31 or fp32_lazy_adam or test_lab_ablation or test_gemm_4w or 8m_8h" > $O/new_tests.txt 2>&1; echo "new tests rc=$?"
grep -E "^\[|passed|failed|FAILED|ERROR|rc=|Error|ViT-L|fp32 operands" $O/new_tests.txt | cut -c1-400 | tail -30
timeout 900 python -m pytest tests/test_gpu_vit.py tests/test_gpu_lab.py -m gpu -q -p no:cacheprovider -x > $O/vit_and_lab_tests.txt 2>&1; echo "vit+lab rc=$?"
tail -3 $O/vit_and_lab_tests.txt
timeout 600 python bench.py > $O/bench_default.log 2>&1; echo "bench rc=$?"; tail -1 $O/bench_default.log > $O/bench_default.json
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05a/bench_default.json").read())
print({k: d.get(k) for k in ("value", "ms_per_step", "value_fp32_fit", "value_fp32", "value_vit_large_k4", "dist_world_size")})
print({k: d["config"].get(k) for k in ("t_extract_s_serial", "t_fit_s_serial")}, d["config"].get("value_vit_large_k4_detail"))
print(d.get("roofline"))
PY
for t in "" "13=2"; do
  for i in 1 2; do
    f=$O/ab_rows_${t:-default}_$i.json
    timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-fp32-fit --no-vit-large --no-probes ${t:+--tune $t} 2>/dev/null | tail -1 > $f
    python -c "import json,sys; d=json.load(open('$f')); print('$f', round(d['value'],3), round(d['ms_per_step'],1))"
  done
done
cd /tmp
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES -d $O/pmc_clock -o clock -- python $R/tools/pmc_clock_target.py > $O/pmc_clock.log 2>&1; echo "pmc rc=$?"
f=$(find $O/pmc_clock -name '*.db' | head -1); [ -n "$f" ] && python $R/tools/pmc_clock_stats.py $f > $O/pmc_clock.txt 2>&1; rm -rf $O/pmc_clock
head -60 $O/pmc_clock.txt | cut -c1-160
