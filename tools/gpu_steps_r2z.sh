out=$1
R=$GRAFT_REPO_ROOT
step() { # name, seconds, command... ; a step that times out ends the call
  local name=$1 secs=$2; shift 2
  timeout -k 10 $secs "$@"; local rc=$?
  echo "$name rc=$rc" | tee -a $R/$out/summary.txt
  if [ $rc -ge 124 ]; then echo "stopping after $name" | tee -a $R/$out/summary.txt; exit 1; fi
}
step "fused head test" 300 bash -c "python -m pytest tests/test_gpu_model.py -q -m gpu -x -k 'fused_frozen' > $out/tests_fh.log 2>&1"
step "model+train tests" 900 bash -c "python -m pytest tests/test_gpu_model.py tests/test_gpu_train.py -q -m gpu -x > $out/tests_a.log 2>&1"
step "bench" 300 bash -c "python bench.py > $out/bench.json 2> $out/bench.err"
step "bench c3" 300 bash -c "python bench.py --points 2048 --profile final --no-cpu-baseline > $out/bench_c3.json 2> $out/bench_c3.err"
step "scan pipeline" 300 bash -c "python tools/bench_scan.py > $out/scan.json 2> $out/scan.err"
cd /tmp
step "prof" 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/prof -o c2 -- python3 $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline
