export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -q -m gpu -x -k "temporal_block or split_independent" 2>&1 | grep -v "MIOpen\|amdgpu.ids" | tail -8
timeout 900 python -m pytest tests/test_gpu_e2e.py -q -m gpu -x -k "model_handle or temporal_blocks or tiled_chunks" 2>&1 | grep -v "MIOpen\|amdgpu.ids" | tail -8
for P in 1 0; do
  VIDTOK_AMD_TBLOCK_PRENORM=$P timeout 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --traffic none --breakdown > gpurun_out/t_bench.json 2> gpurun_out/t_bench.txt
  echo "prenorm=$P: $(python -c "import json; d=json.load(open('gpurun_out/t_bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_step'])")"
  grep "K=  1152  x  9\|K=   768  x  5" gpurun_out/t_bench.txt
done
