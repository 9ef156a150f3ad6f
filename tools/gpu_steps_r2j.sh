out=$1
python tools/graph_concurrency.py > $out/graph_conc.jsonl 2> $out/graph_conc.err; echo "rc=$?"
DEBUG_HIP_FORCE_GRAPH_QUEUES=1 python tools/graph_concurrency.py > $out/graph_conc_q1.jsonl 2>> $out/graph_conc.err; echo "rc=$?"
cat $out/graph_conc.jsonl $out/graph_conc_q1.jsonl
