out=$1
R=$GRAFT_REPO_ROOT
step() { # name, seconds, command... ; a step that times out ends the call
  local name=$1 secs=$2; shift 2
  timeout -k 10 $secs "$@"; local rc=$?
  echo "$name rc=$rc" | tee -a $R/$out/summary.txt
  if [ $rc -ge 124 ]; then echo "stopping after $name" | tee -a $R/$out/summary.txt; exit 1; fi
}
step "ops tests" 900 bash -c "python -m pytest tests/test_gpu_ops.py -q -m gpu -x -k 'max or panel or bf16_storage' > $out/tests_a.log 2>&1"
step "panel probe" 300 bash -c "python tools/panel_probe.py > $out/panel_probe.jsonl 2> $out/panel_probe.err"
step "bench" 300 bash -c "python bench.py --no-cpu-baseline > $out/bench.json 2> $out/bench.err"
step "bench b32n4096" 300 bash -c "python bench.py --points 4096 --no-cpu-baseline > $out/bench_b32n4096.json 2> $out/bench_b32n4096.err"
