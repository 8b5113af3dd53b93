#!/bin/bash
# Every number quoted in DESIGN.md section 5 / 7 and README.md, in the order they appear (run on an MI355X box, repo root).
set -e
python __graft_entry__.py build
python -m pytest tests -q -m "not gpu"
python -m pytest tests -q -m gpu
python __graft_entry__.py smoke
python bench.py                                   # configs[1]: detector step + hot_path / fp32 passes + roofline + cpu_baseline
python bench.py --gpus 2 --no-cpu-baseline        # spawns its own ranks (one per GPU, RCCL)
python bench.py --inflight 2 --no-cpu-baseline    # + the two-frames-in-flight pass
python bench.py --workload cp_lidar --stage train --no-cpu-baseline      # training step of the LiDAR detector
python bench.py --workload cp_fusion --stage train --no-cpu-baseline     # ... with the camera fusion in the graph
python bench.py --workload tf_fusion --no-cpu-baseline                   # configs[2]: TransFusion-L + 3D-DF, bs = 4, bf16
python bench.py --workload vr_fusion --no-cpu-baseline                   # configs[4]: Voxel-RCNN + 3D-DF, KITTI, bs = 8
bash tools/profile_r02.sh && python tools/make_profiles.py gpurun_out/prof_r02 r02   # profiles/r02_* (trace, PMC, calibration)
python tools/bench_trees.py neck 20               # BEV neck (RPN) on the row kernels vs torch/MIOpen
python tools/bench_trees.py head 20               # CenterHead forward + predict; sweep -> boxes
python tools/bench_trees.py tfhead 20             # TransFusionHead forward + get_bboxes (device path vs plain torch)
python tools/ubench/os_probe.py                   # per-layer conv kernel timings (split precision and bf16)
python tools/ubench/sk_probe.py 50                # loader / consumer kernel against the register-gather kernel, values + time
tools/ubench/cu_ingest                            # what a CU takes in per clock by source / path / wave count (hipcc cu_ingest.hip)
tools/ubench/consumer_loop                        # the matrix wave's inner loop in isolation (hipcc consumer_loop.hip)
python tools/ubench/lc_trace.py                   # needs DF3D_HIPCC_FLAGS=-DDF3D_OS_TRACE python 3d-dual-fusion_amd/csrc/build.py
python tools/ubench/ablate_probe.py 50            # needs DF3D_HIPCC_FLAGS=-DDF3D_OS_EXPERIMENTS (DF3D_OS_DBG bits)
bash tools/ubench/conv4_pmc.sh lc                 # L2 hit rate / L1->L2 latency / LDS conflicts / wave-state cycles of conv4
python tools/ubench/xattn_probe.py                # the split-key cross-attention kernel alone
python tools/ubench/topk_probe.py                 # the top-k select of both heads alone
python tools/ubench/fps_probe.py                  # FPS kernel alone (us per iteration at 8 x 24k points)
python tools/ubench/wgrad_probe.py                # filter-gradient kernels on the submanifold layers of a sweep (direct vs LDS-staged)
bash tools/ubench/wgrad_pmc.sh a                  # MFMA busy / wait / LDS / L2 counters of the conv4 filter gradient
bash tools/debug/train_prof.sh                    # kernel table of ONE training step (WL=cp_fusion for the fusion graph) -> profiles/r02_train_*_kernels.txt
python tools/debug/train_phases.py                # host time to queue each phase of the training step vs its wall time
WL=vr_fusion bash tools/debug/wl_prof.sh          # kernel table of an inference workload (cp_fusion | tf_fusion | vr_fusion)
python tools/debug/torch_ops.py                   # ATen ops with device time in one detector step
