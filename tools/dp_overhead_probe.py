#!/usr/bin/env python
"""What does the data-parallel PATH cost before any byte crosses a link?  One rank, RCCL group of one: the step (a) as bench.py runs it
at N = 1 and (b) with FlatGradSync(force=True) in the bench's N > 1 default (three-range phased backward; the first range travels as the
Linear layers' factors -- all-gather + local contraction --, the others by direct exchange / all-reduce; early norm behind the first
range): the phase joins, the wire pack / share sum / unpack kernels, the two collectives per range (self-copies
here) and the events.  T(a) / T(b) is an upper bound of the weak-scaling efficiency the path can reach at any N."""
import os, sys, time
import torch
import torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("MASTER_PORT", "29571"), RANK="0", WORLD_SIZE="1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
from bench import synth_sprites
from lunaris_orion_amd.parallel import FlatGradSync
from lunaris_orion_amd.trainer import VAEStepper
from lunaris_orion_amd.vae import LunarisCoreVAE

B, STEPS = 64, int(os.environ.get("STEPS", "300"))
pool = [synth_sprites(B, s).cuda() for s in range(4)]


def run(sync):
    torch.manual_seed(42)
    m = LunarisCoreVAE(latent_dim=512).to("cuda")
    st = VAEStepper(m, lr=1e-4, min_lr=1e-6, scheduler_t0=10, weight_decay=0.01, max_grad_norm=1.0, recon_weight=1.0, kl_weight=0.1,
                    gradient_accumulation_steps=1, pipeline_optimizer=True, grad_sync=sync)
    for i in range(300):
        st.step(pool[i % 4], batch_idx=i)
    torch.cuda.synchronize()
    if sync is not None:
        sync.reset_timing()
    t = time.perf_counter()
    for i in range(STEPS):
        st.step(pool[i % 4], batch_idx=i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / STEPS
    st.metrics()
    return B / dt, 1e3 * dt


class PhasesOnly:
    """the three-range phased backward without any exchange: what the phase boundaries (side-stream joins) cost on their own"""
    world = 2
    def begin(self, g): pass
    def finish(self): pass
    def __call__(self, g): pass
    def reset_timing(self): pass


for rep in range(2):
    p_ = run(PhasesOnly())
    print(f"phased backward, no exchange: {p_[0]:8.0f} sprites/s {p_[1]:.3f} ms")
    a = run(None)
    s = FlatGradSync(force=True, mode="direct", compress_fp16=True, time_exposed=True)
    b = run(s)
    print(f"single-process step {a[0]:8.0f} sprites/s {a[1]:.3f} ms | data-parallel path, one rank {b[0]:8.0f} sprites/s {b[1]:.3f} ms "
          f"(exposed in finish(): {s.exposed_ms_per_step():.3f} ms, ranges ran as {s.modes_per_phase()}, bytes per range {s.bytes_per_phase()}) | ratio {a[1] / b[1]:.3f}")
dist.destroy_process_group()
