"""time line of one wave's chunk in k_igemm_ws<.., NP = 2> (a -DU2PL_WS_STAMPS build): cycles between every 4th slot.
    python -m u2pl_amd.build_ext --variant stamps -DU2PL_WS_STAMPS
    U2PL_LIB_PATH=u2pl_amd/lib/variants/libu2pl_hip_stamps.so python tools/ws_stamps_h.py"""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from u2pl_amd import _lib
from u2pl_amd._lib import call, query
L = _lib.lib().cdll
DEV = "cuda"
buf = torch.zeros(2 * 2 * 4 * 14 + 1024 * 4 + 64, dtype=torch.int64, device=DEV)
L.u2pl_igemm_ws_set_stamp_buffer(ctypes.c_void_p(buf.data_ptr()))
for (M, K, Nn, batch) in [(10000, 512, 256, 36), (37636, 1024, 256, 1)]:
    x = torch.randn(batch * M * K, device=DEV)
    w = torch.randn(batch * Nn * K, device=DEV) * (K ** -0.5)
    y = torch.empty(batch * M * Nn, device=DEV)
    wsh = torch.empty(query("u2pl_weight_split2h_bytes", Nn, K, batch), dtype=torch.uint8, device=DEV)
    call("u2pl_weight_split2h_f32", w, Nn * K, Nn, K, batch, wsh, torch.empty(64, dtype=torch.uint8, device=DEV))
    amax = torch.zeros(2048, device=DEV)
    call("u2pl_absmax_f32", x, K, batch * M, K, amax, 1)
    for _ in range(3):
        call("u2pl_gemm_batched_wsh_f32", x, K, M * K, amax, wsh, y, Nn, M * Nn, M, K, Nn, batch)
    torch.cuda.synchronize()
    full = buf.cpu()
    t = full[:224].reshape(2, 2, 4, 14)
    print(f"shape M{M} K{K} N{Nn} batch{batch}")
    for b in range(2):
        for p in range(2):
            for c in range(4):
                r = t[b, p, c]
                if r[0] == 0:
                    continue
                d = [int(r[i] - t[b, 0, 0, 0]) for i in (0, 1, 2, 3, 4, 5, 12, 13)]
                print(f"  blk{b} wave{4*p} chunk{8+c}: stamps at slot 0,4,..,20, pre-barrier, post-barrier: {d}  chunk {d[-1]-d[0]}")
    te = full[224 + 4096:224 + 4096 + 40].reshape(2, 4, 5)
    for wv in range(2):
        for n in range(4):
            r = te[wv, n]
            if r[0]:
                print(f"  tile_end wave{5*wv} #{n}: start+{int(r[0]-te[0,0,0])} tail {int(r[1]-r[0])} stores {int(r[2]-r[1])} stats+zero {int(r[3]-r[2])} drain(vmcnt0) {int(r[4]-r[3])}")
    ph = full[224:224 + 4096].reshape(1024, 4)
    ph = ph[ph[:, 0] != 0]
    t0 = int(ph[:, 0].min())
    import statistics as st
    d = lambda a: (int(a.min()), int(st.median(a.tolist())), int(a.max()))
    print("  blocks", len(ph), "start-t0", d(ph[:, 0] - t0), "prologue", d(ph[:, 1] - ph[:, 0]), "mainloop", d(ph[:, 2] - ph[:, 1]),
          "epilogue", d(ph[:, 3] - ph[:, 2]), "end-t0", d(ph[:, 3] - t0))
    buf.zero_()
