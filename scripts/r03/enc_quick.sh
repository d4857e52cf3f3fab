mkdir -p gpurun_out/enc2
python -m pytest tests/test_gpu_bert.py -m gpu -x -q > gpurun_out/enc2/pytest.txt 2>&1; tail -15 gpurun_out/enc2/pytest.txt
python scripts/bench_encoders.py > gpurun_out/enc2/enc.txt 2>&1; grep -E "^(m2v|bert)" gpurun_out/enc2/enc.txt
