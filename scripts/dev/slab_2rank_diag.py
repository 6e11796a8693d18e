import importlib, os, sys, socket
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))

def worker(rank, world, port, iters):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch, torch.distributed as dist, cg_numpy
    pkg = importlib.import_module("conjugate-gradient-pyopencl_amd")
    dmod = importlib.import_module("conjugate-gradient-pyopencl_amd.dist")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0); dev = torch.device("cuda", 0); ctx = pkg.Context(0)
    if os.environ.get("BLOCKDIAG"):
        import scipy.sparse as sp
        i1, x1, d1 = cg_numpy.laplace3d(24, 20, 18)
        A1 = sp.csr_matrix((d1, x1, i1)); A = sp.block_diag([A1, A1], format="csr"); A.sort_indices()
        ip, ix, da = A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data
    else:
        ip, ix, da = cg_numpy.laplace3d(24, 20, 36)
    b = np.linspace(1.0, 2.0, len(ip) - 1)
    n = len(ip) - 1
    ranges = dmod.row_ranges(n, world); rb, re = ranges[rank]; lo, hi = ip[rb], ip[re]
    plan = dmod.build_halo_plan(torch.from_numpy(ix[lo:hi].astype(np.int64)), ranges, rank)
    plan.cols_local = plan.cols_local.to(dev); plan.send_index = plan.send_index.to(dev)
    indptr = torch.from_numpy((ip[rb:re + 1] - lo).astype(np.int32)).to(dev); vals = torch.from_numpy(da[lo:hi]).to(dev)
    lib = pkg._lib.load()
    pkg._lib.check(lib.cgamd_tune(b"index_codes_min_mb", 0)); pkg._lib.check(lib.cgamd_tune(b"dev.resident_lock", 0))
    bl = torch.from_numpy(b[rb:re]).to(dev)
    out = {}
    for name, fl in (("launched", 0), ("slab", 512)):
        s = dmod.DistSolver(ctx, plan, indptr, vals, np.float64, flags=fl, comm="p2p")
        s.set_rhs(bl, None); s.iterate(iters)
        if fl == 512 and plan.n_halo:
            vs = 8; up = lambda b: (b + 255) & ~255
            off0 = 16384 + up(plan.n_halo * vs); body = up((plan.n_local + plan.n_halo) * vs)
            for which in (0, 1):
                buf = np.zeros(plan.n_local + plan.n_halo)
                pkg._lib.check(lib.cgamd_memcpy_d2h(ctx.handle, pkg._lib.ptr(buf), s.mailbox.value + off0 + which * body, buf.nbytes))
                tail = buf[plan.n_local:]
                send = buf[plan.send_index.cpu().numpy()]
                gathered = [None] * world
                dist.all_gather_object(gathered, send)
                peer = plan.peers[0]
                print(f"rank {rank} ds{which}: tail nonzero {np.count_nonzero(tail)}/{len(tail)} max|tail - peer boundary| {np.abs(tail - gathered[peer]).max():.3e} |tail|max {np.abs(tail).max():.3e} body max {np.abs(buf[:plan.n_local]).max():.3e}", flush=True)
        out[name] = (s.x(torch.empty(plan.n_local, dtype=torch.float64, device=dev)).cpu().numpy(), s.history().copy(), lib.cgamd_dist_loop_launches(s.handle), s.p2p_error())
        s.close(); dist.barrier()
    xl, hl, _, _ = out["launched"]; xs, hs, ln, er = out["slab"]
    d = np.abs(xs - xl)
    bad = np.nonzero(d > 1e-9 * np.abs(xl).max())[0]
    print(f"rank {rank}: slab launches {ln} err {er} peers {plan.peers} send {plan.send_counts} recv {plan.recv_counts} n_local {plan.n_local} n_halo {plan.n_halo}; hist slab {hs[:3]} launched {hl[:3]}; bad rows {len(bad)} first {bad[:5]} last {bad[-5:] if len(bad) else []}", flush=True)
    dist.destroy_process_group()

if __name__ == "__main__":
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(worker, args=(2, port, int(sys.argv[1]) if len(sys.argv) > 1 else 16), nprocs=2, join=True)
