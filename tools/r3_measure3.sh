#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py::test_msd_build_equals_structure_major_build tests/test_gpu_fullsize.py::test_swissprot_scale_542000_index_and_planted_motifs tests/test_gpu_configs.py::test_whole_structure_query_at_human_scale -m gpu -x -q 2>&1 | tail -4
FD_BENCH_BACKEND=gloo timeout 1200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline --no-export --build-chunk-blocks 1 > gpurun_out/bench_r3c_g2.json 2> gpurun_out/bench_r3c_g2.err; grep -v "^W0\|amdgpu.ids\|^\*\*\*\|OMP_NUM" gpurun_out/bench_r3c_g2.err | tail -12
python tools/profile_whole_query.py > gpurun_out/whole_query_profile_r3.txt 2>&1; tail -22 gpurun_out/whole_query_profile_r3.txt
bash tools/profile_round3.sh > gpurun_out/r3_profile.log 2>&1; tail -90 gpurun_out/r3_profile.log
