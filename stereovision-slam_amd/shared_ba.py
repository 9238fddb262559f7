"""Shared-map bundle adjustment (BASELINE config 5): ONE local-BA problem whose landmarks are sharded
over the GPUs of a node.  Every rank holds all K poses and a shard of the landmarks with their edges
(a landmark's Schur contribution is self-contained); per LM trial the ranks all-reduce the
reduced camera system — (6K)^2 + 2 (6K) + 1 doubles, 3 781 at K = 10 — over RCCL, every rank solves it
redundantly (same input, same code: bit-identical steps) and back-substitutes its own landmarks.

This file is the LM control flow: g2o's OptimizationAlgorithmLevenberg, statement for statement what the
single-GPU kernel k_local_ba runs on the device (csrc/k_ba.h, the loop after "double lambda"); the
pieces of a trial are the phases of `engine` (Context.sba_phase: phases of k_local_ba_t<1>)."""
import math

import numpy as np


def shared_map_ba(engine, rank, nkf, iters=10, trace=None):
    """engine.phase(p, lam, io) -> io (see include/svslam.h, svslam_sba_phase); rank: dist.Rank.
    Returns (iterations done, final lambda).  The optimised state stays in the engine.
    trace (optional list): receives one record per LM trial — (iteration, lambda, chi2 before, chi2 of the
    trial, rho, accepted) — the layout of svslam_lm_trace, so tests can compare trajectories."""
    n = 6 * nkf
    nio = n * n + 3 * n + 8
    oS, obs, obp, ohd, osc = 0, n * n, n * n + n, n * n + 2 * n, n * n + 3 * n
    lam, ni = 0.0, 2.0
    it_done = 0
    current = None
    for it in range(iters):
        if it == 0:
            io = engine.phase(1, 0.0, np.zeros(nio))
            hd = rank.allreduce(io[ohd:ohd + n], "sum")
            lmax = rank.allreduce(io[osc + 1:osc + 2], "max")[0]
            lam = 1e-5 * max(float(np.abs(hd).max()), float(lmax))
            ni = 2.0
        rho, qmax = 0.0, 0
        while True:
            io = engine.phase(2, lam, np.zeros(nio))
            red = rank.allreduce(np.concatenate([io[oS:ohd], io[osc:osc + 1]]), "sum")     # S, bs, bp | chi2
            if current is None:
                current = float(red[-1])
            S = red[:n * n].reshape(n, n).copy()
            S[np.diag_indices(n)] += lam
            io3 = np.zeros(nio)
            io3[oS:obs] = S.ravel(); io3[obs:ohd] = red[n * n:n * n + 2 * n]
            out = engine.phase(3, lam, io3)
            part = rank.allreduce(np.array([out[osc + 3], out[osc + 5]]), "sum")           # rho denominator (landmarks), chi2
            ok = out[osc + 2] != 0.0
            temp = float(part[1]) if ok else 1.7976931348623157e308
            scale = float(part[0]) + float(out[osc + 4]) + 1e-3
            rho = (current - temp) / scale
            if trace is not None:
                trace.append((float(it), lam, current, temp, rho, 1.0 if (rho > 0 and math.isfinite(temp)) else 0.0))
            if rho > 0 and math.isfinite(temp):
                t = 2 * rho - 1
                alpha = min(1.0 - t * t * t, 2.0 / 3.0)
                lam *= max(1.0 / 3.0, alpha); ni = 2.0; current = temp
            else:
                lam *= ni; ni *= 2
                engine.phase(4, lam, np.zeros(nio))
                if not math.isfinite(lam):
                    break
            qmax += 1
            if not (rho < 0 and qmax < 10):
                break
        it_done += 1
        if qmax == 10 or rho == 0 or not math.isfinite(lam):
            break
    engine.phase(5, lam, np.zeros(nio))
    return it_done, lam


def broadcast_comm_id(rank):
    """rank 0 creates the RCCL unique id, every rank receives it over torch.distributed (whatever backend the
    ranks were launched with); a single rank creates its own"""
    import importlib
    svs = importlib.import_module(__package__)
    if rank.dist is None:
        return svs.sba_comm_unique_id()
    import torch
    t = torch.zeros(128, dtype=torch.uint8)
    if rank.rank == 0:
        t = torch.frombuffer(bytearray(svs.sba_comm_unique_id()), dtype=torch.uint8).clone()
    if rank.device:
        t = t.to(rank.device)
    rank.dist.broadcast(t, src=0)
    return bytes(t.cpu().numpy().tobytes())


def shared_map_ba_native(ctx, rank, iters=10):
    """The product path: the open shard of `ctx` (Context.sba_open) optimised by svslam_sba_solve — LM control flow
    in the library, ncclAllReduce on the device buffer.  Sets the communicator up on first use."""
    if not getattr(ctx, "_sba_comm", False):
        ctx.sba_comm_init(rank.world, rank.rank, broadcast_comm_id(rank))
        ctx._sba_comm = True
    return ctx.sba_solve(iters)


class HipEngine:
    """this rank's shard on its GPU (Context.sba_*)"""

    def __init__(self, ctx, cam_l, ext_l, cam_r, ext_r, poses, pts, okf, olm, ori, ouv, huber_delta=5.991):
        self.ctx = ctx
        self.nio = ctx.sba_open(cam_l, ext_l, cam_r, ext_r, poses, pts, okf, olm, ori, ouv, huber_delta)

    def phase(self, p, lam, io):
        return self.ctx.sba_phase(p, lam, np.ascontiguousarray(io, np.float64))

    def close(self):
        return self.ctx.sba_close()


def shard_by_landmark(nlm, okf, olm, ori, ouv, rank, world):
    """landmark l -> rank l % world; returns (landmark ids of the shard, edge arrays with local landmark numbers)"""
    mine = np.arange(rank, nlm, world)
    local = -np.ones(nlm, np.int64); local[mine] = np.arange(len(mine))
    m = (np.asarray(olm) % world) == rank
    return mine, np.asarray(okf)[m], local[np.asarray(olm)[m]].astype(np.int32), np.asarray(ori)[m], np.asarray(ouv)[m], np.nonzero(m)[0]
