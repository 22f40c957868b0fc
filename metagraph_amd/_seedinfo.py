"""Test hook: per-read seeding intermediates from the device (num_matches, seeds)."""
import ctypes as C

import numpy as np

from . import capi


def fetch(aligner, n):
    L = capi.lib()
    L.mgx_fetch_seed_info.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    info = (C.c_uint32 * (6 * n))()
    ms = C.c_uint32()
    rc = L.mgx_fetch_seed_info(aligner.h, info, None, C.byref(ms))
    assert rc == 0
    seeds = (C.c_uint32 * (n * 2 * ms.value * 4))()
    rc = L.mgx_fetch_seed_info(aligner.h, info, seeds, C.byref(ms))
    assert rc == 0
    out = []
    arr = np.ctypeslib.as_array(seeds).reshape(n, 2, ms.value, 4) if n else None
    for i in range(n):
        nf, nr = info[6 * i + 2], info[6 * i + 3]
        sl = []
        for s, cnt in ((0, nf), (1, nr)):
            sl.append([tuple(int(x) for x in arr[i, s, j]) for j in range(cnt)])
        out.append({"num_matches": (info[6 * i], info[6 * i + 1]), "seeds": tuple(sl),
                    "n_extensions": info[6 * i + 4], "n_columns": info[6 * i + 5]})
    return out
