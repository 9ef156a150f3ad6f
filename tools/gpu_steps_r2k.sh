out=$1
R=$GRAFT_REPO_ROOT
step() { # name, seconds, command... ; a step that times out ends the call
  local name=$1 secs=$2; shift 2
  timeout -k 10 $secs "$@"; local rc=$?
  echo "$name rc=$rc" | tee -a $R/$out/summary.txt
  if [ $rc -ge 124 ]; then echo "stopping after $name" | tee -a $R/$out/summary.txt; exit 1; fi
}
step "ops tests" 600 bash -c "python -m pytest tests/test_gpu_ops.py -q -m gpu -x > $out/tests_ops.log 2>&1"
step "model tests" 700 bash -c "python -m pytest tests/test_gpu_model.py tests/test_gpu_train.py -q -m gpu -x > $out/tests_model.log 2>&1"
step "bench" 300 bash -c "python bench.py > $out/bench.json 2> $out/bench.err"
step "bench f32act" 300 bash -c "python bench.py --precision bf16_f32act --no-cpu-baseline > $out/bench_f32act.json 2> $out/bench_f32act.err"
step "config tests" 500 bash -c "python -m pytest tests/test_gpu_parity_configs.py -q -m gpu > $out/tests_cfg.log 2>&1"
cd /tmp
step "prof" 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/prof -o c2 -- python3 $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline
