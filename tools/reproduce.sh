#!/bin/bash
# Every number quoted in DESIGN.md section 5 / README.md, in the order they appear (run on an MI355X box, repo root).
set -e
python __graft_entry__.py build
python -m pytest tests -q -m "not gpu"
python -m pytest tests -q -m gpu
python __graft_entry__.py smoke
python bench.py                                   # configs[1], two frames in flight, + sequential roofline pass, + cpu_baseline
python bench.py --inflight 1 --no-cpu-baseline    # strictly sequential
python tools/bench_trees.py tf 8                  # TransFusion-L encoder + ACTR, bs=4 (configs[2] shape), fp32 I/O
DF3D_CONV_PRECISION=bf16 python tools/bench_trees.py tf 8    # ... with the bf16 convolution kernels
python tools/bench_trees.py vr 5                  # Voxel-RCNN backbone (MVX + ACTRv2), bs=8 (configs[4] shape)
python tools/bench_trees.py neck 20               # BEV neck (RPN) on the row kernels vs torch/MIOpen
python tools/bench_trees.py head 20               # CenterHead forward + predict; sweep -> boxes
python tools/bench_trees.py tfhead 20             # TransFusionHead forward + get_bboxes (device path vs plain torch)
python tools/bench_trees.py train 10              # backbone training step (forward + backward)
python tools/ubench/xattn_probe.py                # the split-key cross-attention kernel alone
python tools/ubench/topk_probe.py                 # the top-k select of both heads alone
python tools/ubench/os_probe.py                   # per-layer conv kernel timings (split precision and bf16)
# profiles/: rocprofv3 --kernel-trace --stats and --pmc FETCH_SIZE / WRITE_SIZE passes, distilled by tools/make_profiles.py
