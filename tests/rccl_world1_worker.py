"""Child process of tests/test_gpu_multirank.py::test_rccl_world_size_one_beside_a_live_pipeline: RCCL (backend "nccl") initialised
through the PRODUCT's init (spinrelax_amd.dist.init_group: device bound, communicator created for that device) with a world of
one rank, in the same process as a live Context + GroupedPipeline -- shared HIP runtime, hardware queues (GPU_MAX_HW_QUEUES),
signal memory.  One all-gather of a device tensor queued behind a pipeline step, the product's barrier, destroy, exit 0."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault('GPU_MAX_HW_QUEUES', '10')
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')


def main():
    import numpy as np
    import torch
    import torch.distributed as dist
    from spinrelax_amd import dist as srdist
    from spinrelax_amd import synth
    from spinrelax_amd.hip import Context
    from spinrelax_amd.pipeline import GroupedPipeline

    assert int(os.environ['WORLD_SIZE']) == 1 and int(os.environ['RANK']) == 0
    srdist.init_group('nccl')
    assert dist.is_initialized() and dist.get_backend() == 'nccl' and dist.get_world_size() == 1
    dev = torch.device('cuda', srdist.local_device())
    assert torch.cuda.current_device() == dev.index
    s = synth.config_shapes(1)
    vecs = torch.from_numpy(synth.synth_config(1)).to(dev)
    ctx = Context(dev.index)
    pipe = GroupedPipeline(ctx, dev, s['frames'], s['V'], s['R'], s['F'], s['dt'], group=2, q_rot=synth.Q_EXT, Diso=synth.DISO,
                           aniso=synth.DANI, field_MHz=(synth.FIELD_MHZ,), zeta=synth.ZETA, stream=torch.cuda.Stream(device=dev))
    out = {}

    def gather(grp):
        # a device-side consumer of the finished group, as bench.py's: RCCL all-gather behind the group's last launch
        gs = out.setdefault('stream', torch.cuda.Stream(device=dev))
        with torch.cuda.stream(gs):
            gs.wait_event(grp.done)
            parts = [torch.empty_like(grp.relax)]
            dist.all_gather(parts, grp.relax.contiguous())
            out['relax'] = parts[0].clone()
            ev = torch.cuda.Event()
            ev.record(gs)
        return ev

    with torch.cuda.stream(pipe.main):
        pipe.prime(vecs)
        pipe.run(vecs, 4, None, None, gather)
    torch.cuda.synchronize()
    table = pipe.relax_out.copy()
    got = out['relax'].cpu().numpy()
    assert got.shape[1] == 2 * s['V'] and np.array_equal(got[:, -s['V']:], table, equal_nan=True), (got.shape, table.shape)
    # the library's gather helper on a device tensor (world 1: identity) and a plain collective on the default stream
    t = torch.arange(12, device=dev, dtype=torch.float64).reshape(3, 4)
    assert torch.equal(srdist.gather_vector_axis(t, 4, 1), t)
    parts = [torch.empty_like(t)]
    dist.all_gather(parts, t)
    dist.all_reduce(t)
    torch.cuda.synchronize()
    assert torch.equal(parts[0], t)
    pipe.close()
    ctx.close()
    srdist.finish()                      # barrier(device_ids=[...]) + destroy_process_group
    assert not dist.is_initialized()
    print('RCCL_WORLD1_OK')


if __name__ == '__main__':
    main()
