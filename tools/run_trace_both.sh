bash tools/run_trace_n1.sh 1
bash tools/run_trace_n1.sh 8
