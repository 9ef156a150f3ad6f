out=$1
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py tests/test_gpu_train.py -q -m gpu -x > $out/tests_a.log 2>&1; echo "ops+model+train tests rc=$?" | tee -a $out/summary.txt
timeout -k 10 300 python bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc=$?" | tee -a $out/summary.txt
timeout -k 10 300 python tools/panel_probe.py > $out/panel_probe.jsonl 2> $out/panel_probe.err; echo "probe rc=$?" | tee -a $out/summary.txt
timeout -k 10 300 python tools/resolve_probe.py > $out/resolve_probe.jsonl 2> $out/resolve_probe.err; echo "resolve probe rc=$?" | tee -a $out/summary.txt
cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 5 --no-cpu-baseline > $GRAFT_REPO_ROOT/$out/prof_bench.json 2> $GRAFT_REPO_ROOT/$out/prof.err; echo "prof rc=$?" | tee -a $GRAFT_REPO_ROOT/$out/summary.txt
