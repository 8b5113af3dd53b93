"""The N>1 path on CPU: two processes over gloo (127.0.0.1) exercise frame sharding, the
barrier/max-over-ranks timing protocol of bench.py and the loss-scalar reductions."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import os, sys, time, json
    sys.path.insert(0, os.path.join(%r, "3d-dual-fusion_amd"))
    import torch
    from dualfusion import dist as D
    rank, local, world = D.init_from_env("gloo")
    assert world == 2 and D.is_dist()
    frames = D.frame_shard(7, rank, world)
    D.barrier()
    t0 = time.perf_counter()
    time.sleep(0.05 * (rank + 1))              # rank 1 is the slow one
    D.barrier()
    el = D.max_over_ranks(time.perf_counter() - t0)
    losses = {"loss": torch.tensor(1.0 + rank), "hm_loss": torch.tensor(10.0 * (rank + 1))}
    red = D.reduce_dict(losses)
    allr = D.all_reduce_value(torch.tensor([float(rank + 1)]), "sum", average=True)
    with open(os.path.join(os.environ["DF3D_TEST_OUT"], "rank%%d.json" %% rank), "w") as f:
        json.dump({"rank": rank, "frames": frames, "elapsed": el, "loss": float(red["loss"]),
                   "hm": float(red["hm_loss"]), "avg": float(allr)}, f)
    D.barrier()
    torch.distributed.destroy_process_group()
""") % ROOT


def test_two_rank_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = 29500 + (os.getpid() % 2000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), str(script)]
    env = dict(os.environ, OMP_NUM_THREADS="1", DF3D_TEST_OUT=str(tmp_path))
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    import json
    res = {}
    for r in (0, 1):       # one file per rank: stdout of the two processes can interleave
        with open(os.path.join(str(tmp_path), "rank%d.json" % r)) as f:
            res[r] = json.load(f)
    assert set(res) == {0, 1}
    assert res[0]["frames"] == [0, 2, 4, 6] and res[1]["frames"] == [1, 3, 5]      # disjoint, complete
    assert abs(res[0]["elapsed"] - res[1]["elapsed"]) < 1e-9 and res[0]["elapsed"] >= 0.1   # MAX over ranks
    assert abs(res[0]["loss"] - 1.5) < 1e-6 and abs(res[0]["hm"] - 15.0) < 1e-6    # rank 0 holds the average
    assert abs(res[0]["avg"] - 1.5) < 1e-6 and abs(res[1]["avg"] - 1.5) < 1e-6


def test_bench_self_spawn_two_ranks_gloo():
    """`python bench.py --gpus 2` ALONE (no torch.distributed.run around it) starts its own two ranks through
    dualfusion.dist.launch_ranks -- the same spawn code the GPU run uses -- and rank 0 prints one JSON line with
    n_gpus = 2 and the rank-averaged loss scalars (protocol workload: no device work, gloo instead of RCCL)."""
    import json
    env = dict(os.environ, OMP_NUM_THREADS="1")
    env.pop("WORLD_SIZE", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "protocol",
                          "--backend", "gloo", "--steps", "4", "--warmup", "1"], capture_output=True, text=True,
                         timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["world_size"] == 2 and res["collective_backend"] == "gloo"
    assert res["steps"] == 4 and res["warmup"] == 1 and res["scaling"] == "weak"
    assert res["config"]["global_batch"] == 2
    assert abs(res["reduced_losses"]["loss"][0] - 1.5) < 1e-6 and abs(res["reduced_losses"]["hm_loss"][0] - 15.0) < 1e-6
    assert res["ms_per_step"] >= 2.0 and abs(res["value"] - 2 * 1e3 / res["ms_per_step"]) < 0.5


def test_bench_eight_ranks_gloo_driver_launch():
    """The launch line the driver uses at the target width -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node 8
    --master-addr 127.0.0.1 --master-port P bench.py --gpus 8 ...` -- with the protocol workload over gloo: eight ranks
    rendezvous, barrier, time the steps (MAX over ranks), reduce the loss scalars to rank 0, ONE JSON line, n_gpus = 8."""
    import json
    env = dict(os.environ, OMP_NUM_THREADS="1")
    env.pop("WORLD_SIZE", None)
    port = 31500 + (os.getpid() % 2000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "8", "--workload", "protocol",
           "--backend", "gloo", "--steps", "4", "--warmup", "1"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    res = json.loads(lines[0])
    assert res["n_gpus"] == 8 and res["world_size"] == 8 and res["collective_backend"] == "gloo"
    assert res["config"]["global_batch"] == 8 and res["scaling"] == "weak"
    # loss = 1 + rank averaged over 8 ranks = 4.5; hm_loss = 10 (1 + rank) -> 45
    assert abs(res["reduced_losses"]["loss"][0] - 4.5) < 1e-6 and abs(res["reduced_losses"]["hm_loss"][0] - 45.0) < 1e-5
    assert abs(res["value"] - 8 * 1e3 / res["ms_per_step"]) < 2.0


def test_init_from_env_refuses_a_private_group_when_world_size_says_more(tmp_path):
    """WORLD_SIZE > 1 without MASTER_PORT must not fall back to a private one-rank group (the ranks would
    shard the frames by WORLD_SIZE and reduce over a world of one): it raises."""
    script = tmp_path / "w.py"
    script.write_text(textwrap.dedent("""
        import os, sys
        sys.path.insert(0, os.path.join(%r, "3d-dual-fusion_amd"))
        from dualfusion import dist as D
        try:
            D.init_from_env("gloo")
        except RuntimeError as e:
            assert "MASTER_PORT" in str(e), e
            print("REFUSED")
        else:
            print("ACCEPTED")
    """) % ROOT)
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1")
    env.pop("MASTER_PORT", None)
    out = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=120, env=env)
    assert out.returncode == 0 and "REFUSED" in out.stdout, out.stdout + out.stderr[-1500:]


def test_reduce_dict_mixed_shapes_two_ranks(tmp_path):
    """reduce_dict with per-task vectors and a matrix in one dict (CenterHead.loss_device's layout): one collective."""
    worker = textwrap.dedent("""
        import os, sys, json
        sys.path.insert(0, os.path.join(%r, "3d-dual-fusion_amd"))
        import torch
        from dualfusion import dist as D
        rank, local, world = D.init_from_env("gloo")
        d = {"loss": torch.arange(6.0) + rank, "elem": torch.ones(6, 10) * (rank + 1), "n": torch.tensor(4.0 * rank)}
        r = D.reduce_dict(d)
        if rank == 0:
            json.dump({"loss": r["loss"].tolist(), "elem": float(r["elem"].mean()), "shape": list(r["elem"].shape),
                       "n": float(r["n"])}, open(os.path.join(os.environ["DF3D_TEST_OUT"], "r.json"), "w"))
        D.barrier()
        torch.distributed.destroy_process_group()
    """) % ROOT
    script = tmp_path / "w.py"
    script.write_text(worker)
    sys.path.insert(0, os.path.join(ROOT, "3d-dual-fusion_amd"))
    from dualfusion import dist as D
    rc = D.launch_ranks(2, str(script), [], env=dict(os.environ, OMP_NUM_THREADS="1", DF3D_TEST_OUT=str(tmp_path)),
                        timeout=300)
    assert rc == 0
    import json
    r = json.load(open(os.path.join(str(tmp_path), "r.json")))
    assert r["loss"] == [0.5, 1.5, 2.5, 3.5, 4.5, 5.5] and abs(r["elem"] - 1.5) < 1e-6 and r["shape"] == [6, 10]
    assert abs(r["n"] - 2.0) < 1e-6


def test_gradient_reduction_two_ranks(tmp_path):
    """allreduce_grads (the reference's synchronous coalesced average) and GradBucketReducer (bucketed, overlapped with
    backward, gradients living inside the buckets) against the analytic average over two ranks, incl. a parameter that
    gets no gradient and a second step after zero_grad()."""
    worker = textwrap.dedent("""
        import os, sys, json
        sys.path.insert(0, os.path.join(%r, "3d-dual-fusion_amd"))
        import torch
        from dualfusion import dist as D
        rank, local, world = D.init_from_env("gloo")
        torch.manual_seed(0)
        net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.ReLU(), torch.nn.Linear(5, 3))
        unused = torch.nn.Parameter(torch.ones(4))
        params = list(net.parameters()) + [unused]
        x = torch.arange(12.0).view(2, 6) * (rank + 1) / 10
        def local_grads():
            for p in params:
                p.grad = None
            net(x).square().sum().backward()
            return [None if p.grad is None else p.grad.clone() for p in params]
        mine = local_grads()
        # analytic average: gather both ranks' local gradients
        want = []
        for g in mine[:-1]:
            both = [torch.zeros_like(g) for _ in range(world)]
            torch.distributed.all_gather(both, g)
            want.append(sum(both) / world)
        # (a) the reference's synchronous path, coalesced / bucketed / per tensor
        errs = []
        for kw in (dict(coalesce=True), dict(coalesce=True, bucket_size_mb=1), dict(coalesce=False)):
            local_grads()
            D.allreduce_grads(params, **kw)
            errs.append(max(float((p.grad - w).abs().max()) for p, w in zip(params[:-1], want)))
        # (b) the overlapped reducer: gradients are views of two buckets (tiny bucket size forces a split)
        red = D.GradBucketReducer(params, bucket_mb=0.0002)
        nb = len(red.buckets)
        for step in range(2):
            red.zero_grad()
            net(x).square().sum().backward()
            red.finish()
            errs.append(max(float((p.grad - w).abs().max()) for p, w in zip(params[:-1], want)))
        ok_unused = bool((unused.grad == 0).all())
        views = all(p.grad.data_ptr() == red._view(p._df3d_bucket, p).data_ptr() for p in params)
        with open(os.path.join(os.environ["DF3D_TEST_OUT"], "g%%d.json" %% rank), "w") as f:
            json.dump({"errs": errs, "buckets": nb, "unused_zero": ok_unused, "views": views}, f)
        D.barrier()
        torch.distributed.destroy_process_group()
    """) % ROOT
    script = tmp_path / "w.py"
    script.write_text(worker)
    sys.path.insert(0, os.path.join(ROOT, "3d-dual-fusion_amd"))
    from dualfusion import dist as D
    rc = D.launch_ranks(2, str(script), [], env=dict(os.environ, OMP_NUM_THREADS="1", DF3D_TEST_OUT=str(tmp_path)),
                        timeout=300)
    assert rc == 0
    import json
    for r in (0, 1):
        res = json.load(open(os.path.join(str(tmp_path), "g%d.json" % r)))
        assert max(res["errs"]) < 1e-5, res
        assert res["buckets"] >= 2 and res["unused_zero"] and res["views"], res


def test_bucket_adamw_two_ranks_equals_one_rank_on_the_union_batch(tmp_path):
    """GradBucketReducer + BucketAdamW (parameters and AdamW state as one flat buffer per bucket) over two ranks, each with half
    of a batch: after three steps (with gradient clipping over the flat buffers) both ranks hold the SAME parameters, equal to
    those of one process that ran torch.optim.AdamW on the whole batch (mean loss)."""
    worker = textwrap.dedent("""
        import os, sys, json, copy
        sys.path.insert(0, os.path.join(%r, "3d-dual-fusion_amd"))
        import torch
        from dualfusion import dist as D
        rank, local, world = D.init_from_env("gloo")
        torch.manual_seed(0)
        net = torch.nn.Sequential(torch.nn.Linear(6, 9), torch.nn.ReLU(), torch.nn.Linear(9, 4), torch.nn.LayerNorm(4))
        ref = copy.deepcopy(net)
        red = D.GradBucketReducer(list(net.parameters()), bucket_mb=0.0002)
        opt = D.BucketAdamW(red, lr=1e-2, weight_decay=0.05)
        ropt = torch.optim.AdamW(ref.parameters(), lr=1e-2, weight_decay=0.05)
        gen = torch.Generator().manual_seed(1)
        for step in range(3):
            x = torch.randn((8, 6), generator=gen)                 # the same on both ranks; rank r takes rows [4 r, 4 r + 4)
            red.zero_grad()
            net(x[4 * rank:4 * rank + 4]).square().mean().backward()
            red.finish()
            opt.clip_grad_norm(max_norm=0.5)
            opt.step()
            ropt.zero_grad(set_to_none=True)
            ref(x).square().mean().backward()
            torch.nn.utils.clip_grad_norm_(ref.parameters(), max_norm=0.5)
            ropt.step()
        err = max(float((a - b).abs().max()) for a, b in zip(net.parameters(), ref.parameters()))
        flat = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
        both = [torch.zeros_like(flat) for _ in range(world)]
        torch.distributed.all_gather(both, flat)
        with open(os.path.join(os.environ["DF3D_TEST_OUT"], "a%%d.json" %% rank), "w") as f:
            json.dump({"err": err, "same": bool(torch.equal(both[0], both[1])), "buckets": len(red.buckets)}, f)
        D.barrier()
        torch.distributed.destroy_process_group()
    """) % ROOT
    script = tmp_path / "w.py"
    script.write_text(worker)
    sys.path.insert(0, os.path.join(ROOT, "3d-dual-fusion_amd"))
    from dualfusion import dist as D
    rc = D.launch_ranks(2, str(script), [], env=dict(os.environ, OMP_NUM_THREADS="1", DF3D_TEST_OUT=str(tmp_path)),
                        timeout=300)
    assert rc == 0
    import json
    for r in (0, 1):
        res = json.load(open(os.path.join(str(tmp_path), "a%d.json" % r)))
        assert res["err"] < 2e-5 and res["same"] and res["buckets"] >= 2, res


def test_gradient_buckets_rank_dependent_unused_parameters(tmp_path):
    """the set of parameters without a gradient differs between the ranks (a data-dependent branch: rank 0
    skips the LAST layer's bucket, rank 1 the FIRST layer's).  Buckets are launched strictly in index order, so both ranks
    issue the same sequence of collectives and the averaged gradients are right (an as-soon-as-complete launch order
    pairs buffers of different buckets across ranks: wrong sums or a hang)."""
    worker = textwrap.dedent("""
        import os, sys, json
        sys.path.insert(0, os.path.join(%r, "3d-dual-fusion_amd"))
        import torch
        from dualfusion import dist as D
        rank, local, world = D.init_from_env("gloo")
        torch.manual_seed(0)
        a, b, c = torch.nn.Linear(8, 8), torch.nn.Linear(8, 8), torch.nn.Linear(8, 8)
        params = list(a.parameters()) + list(b.parameters()) + list(c.parameters())
        red = D.GradBucketReducer(params, bucket_mb=0.0002)            # one bucket per tensor or so
        order = []
        real = torch.distributed.all_reduce
        def spy(t, *args, **kw):
            order.append(int(t.numel()))
            return real(t, *args, **kw)
        torch.distributed.all_reduce = spy
        x = torch.arange(16.0).view(2, 8) / 10
        def run():
            red.zero_grad()
            h = a(x)
            y = b(h) if rank == 0 else c(h)                 # rank 0 never touches c, rank 1 never touches b
            y.square().sum().backward()
            red.finish()
        run(); run()
        torch.distributed.all_reduce = real
        # expected: average of (rank-0 gradient, rank-1 gradient), zeros where a rank skipped the layer
        def local(r):
            for p in params:
                p.grad = None
            h = a(x)
            (b(h) if r == 0 else c(h)).square().sum().backward()
            return [torch.zeros_like(p) if p.grad is None else p.grad.clone() for p in params]
        got = [p.grad.clone() for p in params]
        red.remove()
        want = [(g0 + g1) / 2 for g0, g1 in zip(local(0), local(1))]
        err = max(float((g - w).abs().max()) for g, w in zip(got, want))
        sizes = [int(bk["flat"].numel()) for bk in red.buckets]
        with open(os.path.join(os.environ["DF3D_TEST_OUT"], "u%%d.json" %% rank), "w") as f:
            json.dump({"err": err, "order": order, "sizes": sizes}, f)
        D.barrier()
        torch.distributed.destroy_process_group()
    """) % ROOT
    script = tmp_path / "w.py"
    script.write_text(worker)
    sys.path.insert(0, os.path.join(ROOT, "3d-dual-fusion_amd"))
    from dualfusion import dist as D
    rc = D.launch_ranks(2, str(script), [], env=dict(os.environ, OMP_NUM_THREADS="1", DF3D_TEST_OUT=str(tmp_path)),
                        timeout=300)
    assert rc == 0
    import json
    res = [json.load(open(os.path.join(str(tmp_path), "u%d.json" % r))) for r in (0, 1)]
    assert res[0]["order"] == res[1]["order"] == res[0]["sizes"] * 2 and len(res[0]["sizes"]) >= 3, res
    assert max(r["err"] for r in res) < 1e-6, res


def test_all_gather_of_detections_two_ranks(tmp_path):
    """The evaluation loop's collective (CP/det3d/torchie/trainer/utils.py:114-157, trainer.py:471): `all_gather` of
    picklable per-rank detections, and `gather_detections` -- the same merge for the fixed-capacity tensors of the device
    tails, one packed buffer per rank, 7 frames sharded 4 + 3."""
    worker = textwrap.dedent("""
        import os, sys, json
        sys.path.insert(0, os.path.join(%r, "3d-dual-fusion_amd"))
        import torch
        from dualfusion import dist as D
        rank, local, world = D.init_from_env("gloo")
        frames = D.frame_shard(7, rank, world)
        dets = {"tok%%d" %% f: {"scores": torch.full((f + 1,), float(f)), "meta": "r%%d" %% rank} for f in frames}
        allp = D.all_gather(dets)
        merged = {}
        for p in allp:
            merged.update(p)
        cap, dim = 5, 9
        g = torch.Generator().manual_seed(100)
        every = torch.randn(7, cap, dim, generator=g)
        counts = torch.tensor([(f * 3) %% (cap + 1) for f in frames])
        boxes, scores = every[frames], every[frames][:, :, 0] * 2
        labels = torch.arange(cap).repeat(len(frames), 1) + torch.tensor(frames)[:, None]
        got = D.gather_detections(frames, boxes, scores, labels, counts)
        ok = sorted(got) == list(range(7))
        for f in range(7):
            k = (f * 3) %% (cap + 1)
            ok = ok and got[f]["box3d_lidar"].shape == (k, dim) and torch.equal(got[f]["box3d_lidar"], every[f, :k])
            ok = ok and torch.equal(got[f]["scores"], every[f, :k, 0] * 2) and got[f]["label_preds"].tolist() == [f + i for i in range(k)]
        with open(os.path.join(os.environ["DF3D_TEST_OUT"], "rank%%d.json" %% rank), "w") as fh:
            json.dump({"keys": sorted(merged), "lens": [int(merged[k]["scores"].numel()) for k in sorted(merged)],
                       "meta": [merged[k]["meta"] for k in sorted(merged)], "ok": bool(ok)}, fh)
        D.barrier()
        torch.distributed.destroy_process_group()
    """) % ROOT
    script = tmp_path / "worker.py"
    script.write_text(worker)
    port = 29500 + ((os.getpid() + 991) % 2000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), str(script)]
    env = dict(os.environ, OMP_NUM_THREADS="1", DF3D_TEST_OUT=str(tmp_path))
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    import json
    for r in (0, 1):
        with open(os.path.join(str(tmp_path), "rank%d.json" % r)) as f:
            res = json.load(f)
        assert res["keys"] == ["tok%d" % f for f in range(7)] and res["lens"] == [f + 1 for f in range(7)]
        assert res["meta"] == ["r0", "r1", "r0", "r1", "r0", "r1", "r0"] and res["ok"]


def test_transfusion_training_step_two_ranks_equals_the_union_of_their_batches(tmp_path):
    """BASELINE configs[3] (TransFusion-L + 3D-DF sharded over the ranks) on two gloo ranks: `TransFusionDetector.
    training_step` with the bucketed reducer, gradient clipping and AdamW -- rank r holds sample r.  The reduced gradients
    (and the weights after the optimizer step) must equal what ONE process gets from the union of the two batches the way
    data parallelism defines it: the mean over the ranks of each rank's own loss (every rank normalises by its own number
    of matched boxes, as the reference does under MMDistributedDataParallel).  BatchNorm runs on its running statistics
    here (per-rank batch statistics are not a function of the union; the reference does not synchronise them either).
    No GPU: the detector's own modules in float64 with the kernel leaves replaced by tests/f64_reference.py."""
    worker = textwrap.dedent("""
        import copy, json, os, sys
        ROOT = %r
        sys.path[:0] = [ROOT, os.path.join(ROOT, "3d-dual-fusion_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
        import numpy as np, torch
        import f64_reference as fr
        from oracle import oracle as orc
        from dualfusion import dist as D
        from dualfusion.transfusion import parse_losses, clip_grads
        rank, local, world = D.init_from_env("gloo")
        torch.manual_seed(0)
        det = fr.small_transfusion_detector(num_proposals=16).double().train()
        for m in det.modules():
            if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
                m.eval()
        points, img, metas, gts, labels = fr.small_inputs(world, seed=3)
        img = torch.from_numpy(img).double()

        def sample(b):
            v, c, n = orc.hard_voxelize(points[b][:4000], fr.SMALL_VOXEL, fr.SMALL_RANGE, 10, 20000)
            coors = np.concatenate([np.zeros((len(c), 1), np.int32), c], 1)
            return dict(voxels=(torch.from_numpy(orc.mean_vfe(v, n)).double(), torch.from_numpy(coors)),
                        img_feats=[img[b * 6:(b + 1) * 6]], img_metas=[dict(metas[b])],
                        gt_bboxes_3d=[torch.from_numpy(gts[b]).double()], gt_labels_3d=[torch.from_numpy(labels[b])])

        clip = dict(max_norm=0.1, norm_type=2)
        ref = copy.deepcopy(det)
        params = [p for p in det.parameters() if p.requires_grad]
        reducer = D.GradBucketReducer(params, bucket_mb=4.0)
        opt = torch.optim.AdamW(params, lr=1e-3, weight_decay=0.01)
        with fr.patched():
            s = sample(rank)
            loss, logs = det.training_step(None, s["img_feats"], s["img_metas"], s["gt_bboxes_3d"], s["gt_labels_3d"],
                                           reducer=reducer, optimizer=None, grad_clip=None, voxels=s["voxels"])
            reduced = [p.grad.clone() for p in params]
            norm = clip_grads(params, **clip)
            opt.step()
            res = dict(buckets=len(reducer.buckets), loss=float(loss), grad_norm=float(norm))
            if rank == 0:
                rparams = [p for p in ref.parameters() if p.requires_grad]
                ropt = torch.optim.AdamW(rparams, lr=1e-3, weight_decay=0.01)
                losses = []
                for b in range(world):
                    s = sample(b)
                    l, _ = parse_losses(ref.forward_train_voxels(s["voxels"][0], s["voxels"][1], 1, s["img_feats"], s["img_metas"],
                                                                 s["gt_bboxes_3d"], s["gt_labels_3d"]))
                    (l / world).backward()
                    losses.append(float(l))
                gerr = max(float((g - (p.grad if p.grad is not None else torch.zeros_like(p))).abs().max())
                           / max(1e-12, float(g.abs().max()), float(p.grad.abs().max()) if p.grad is not None else 0.0)
                           for g, p in zip(reduced, rparams))
                rnorm = clip_grads(rparams, **clip)
                ropt.step()
                # (a parameter the configuration never reaches holds a zero gradient in the reducer's bucket and none in the
                # plain run: AdamW's weight decay moves the former by lr * wd * |p| and skips the latter)
                werr = max(float((p - q).abs().max()) for p, q in zip(params, rparams) if q.grad is not None)
                decay = max([float(((p - q).abs() - 1e-5 * q.abs() * (1 + 1e-9)).max()) for p, q in zip(params, rparams)
                             if q.grad is None] + [0.0])
                res.update(gerr=gerr, werr=werr, decay=decay, ref_norm=float(rnorm), ref_losses=losses,
                           nonzero=sum(int(float(g.abs().max()) > 0) for g in reduced), nparams=len(params))
        with open(os.path.join(os.environ["DF3D_TEST_OUT"], "tf%%d.json" %% rank), "w") as f:
            json.dump(res, f)
        D.barrier()
        torch.distributed.destroy_process_group()
    """) % ROOT
    script = tmp_path / "tfw.py"
    script.write_text(worker)
    sys.path.insert(0, os.path.join(ROOT, "3d-dual-fusion_amd"))
    from dualfusion import dist as D
    rc = D.launch_ranks(2, str(script), [], env=dict(os.environ, OMP_NUM_THREADS="4", DF3D_TEST_OUT=str(tmp_path)),
                        timeout=900)
    assert rc == 0
    import json
    r0 = json.load(open(os.path.join(str(tmp_path), "tf0.json")))
    r1 = json.load(open(os.path.join(str(tmp_path), "tf1.json")))
    assert r0["buckets"] >= 2 and r0["nonzero"] >= r0["nparams"] - 3, r0
    assert r0["gerr"] <= 1e-9 and r0["werr"] <= 1e-12 and r0["decay"] <= 1e-15, r0      # float64 on both sides
    assert abs(r0["grad_norm"] - r0["ref_norm"]) <= 1e-9 * max(1.0, r0["ref_norm"]), r0
    assert abs(r0["grad_norm"] - r1["grad_norm"]) <= 1e-12 * max(1.0, r0["grad_norm"])      # both ranks hold the same gradients
    assert abs(r0["loss"] - r0["ref_losses"][0]) <= 1e-9 and abs(r1["loss"] - r0["ref_losses"][1]) <= 1e-9, (r0, r1)
