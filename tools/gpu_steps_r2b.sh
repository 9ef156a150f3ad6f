out=$1
timeout -k 10 900 python -m pytest tests/test_gpu_parity_configs.py tests/test_gpu_train.py -q -m gpu --durations=8 > $out/tests_new.log 2>&1; echo "new tests rc=$?" | tee -a $out/summary.txt
timeout -k 10 600 python -m pytest tests -q -m gpu -x --deselect tests/test_gpu_parity_configs.py --deselect tests/test_gpu_train.py > $out/tests_old.log 2>&1; echo "old tests rc=$?" | tee -a $out/summary.txt
timeout -k 10 300 python tools/bench_scan.py > $out/scan.json 2> $out/scan.err; echo "scan rc=$?" | tee -a $out/summary.txt
