cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r06t2
timeout 2400 python -m pytest tests -q -m gpu -x > gpurun_out/r06t2/pytest_gpu.txt 2>&1; echo "full gpu suite rc=$?"; tail -4 gpurun_out/r06t2/pytest_gpu.txt
PBSO_ENGINE_OPTS=submit_thread=1 timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_time_chunks.py tests/test_gpu_direct_hits.py tests/test_gpu_listener_mix_edges.py tests/test_gpu_ffat_shared.py tests/test_gpu_large_row_counts.py -q -m gpu --deselect "tests/test_gpu_time_chunks.py::test_stream_hand_over_by_value_equals_the_event_path" > gpurun_out/r06t2/pytest_thread.txt 2>&1; echo "suites with the thread rc=$?"; tail -4 gpurun_out/r06t2/pytest_thread.txt
bash scripts/debug/r06_quick_configs.sh
