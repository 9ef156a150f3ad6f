out=$1
R=$GRAFT_REPO_ROOT
step() { # name, seconds, command... ; a step that times out ends the call
  local name=$1 secs=$2; shift 2
  local t0=$(date +%s)
  timeout -k 10 $secs "$@"; local rc=$?
  echo "$name rc=$rc ($(( $(date +%s) - t0 )) s)" | tee -a $R/$out/summary.txt
  if [ $rc -ge 124 ]; then echo "stopping after $name" | tee -a $R/$out/summary.txt; exit 1; fi
}
step "traj" 300 bash -c "python tools/train_traj.py > $out/traj.jsonl 2> $out/traj.err"
step "panel tests" 400 bash -c "python -m pytest tests/test_gpu_ops.py -q -m gpu -x -k 'max or panel' > $out/tests_panel.log 2>&1"
step "panel stamps" 200 bash -c "python tools/panel_stamps.py > $out/panel_stamps.jsonl 2> $out/panel_stamps.err"
step "bench" 300 bash -c "python bench.py --no-cpu-baseline > $out/bench.json 2> $out/bench.err"
step "bench n4096" 300 bash -c "python bench.py --steps 100 --warmup 10 --no-cpu-baseline --points 4096 > $out/bench_b32n4096.json 2> $out/bench_b32n4096.err"
