"""Calls every entry point of the C ABI with arguments it must refuse -- null pointers, zero / negative sizes, shapes beyond
the documented limits -- and prints one line per call.  Every call has to come back with VX_EINVAL (-1) BEFORE the library
touches the HIP runtime: the process has no GPU (run under AddressSanitizer + UBSan by tests/test_abi_einval.py, where an
out-of-bounds read of an argument array or a signed overflow in a size computation ends the process instead)."""
import ctypes
import sys

from vipsy_amd import _hip

SKIP = ("vx_abi_version", "vx_build_info", "vx_prof_enable", "vx_prof_count", "vx_prof_read", "vx_prof_units",
        "vx_sum_workspace_floats")
HOST = (ctypes.c_float * 4096)()                      # a non-null HOST buffer: nothing may be read through it either


def irt_cfg(kind):
    if kind == "zero":
        return _hip.IrtCfg()
    if kind == "plausible":                           # the headline shape
        return _hip.IrtCfg(2, 100, 500, 64, 1.0, 1.0, 1234, 0, 0)
    if kind == "too_big":                             # beyond every documented limit
        return _hip.IrtCfg(2, 4000, 70000, 4096, 1.0, 1.0, 1234, 0, 0)
    if kind == "negative":
        return _hip.IrtCfg(7, -3, -5, -64, 1.0, 1.0, 0, 0, 0)
    raise KeyError(kind)


def hd_cfg(kind):
    c = _hip.HoDinaCfg()
    if kind == "plausible":
        c.K, c.J, c.H, c.scale = 8, 30, 64, 1.0
    elif kind == "too_big":
        c.K, c.J, c.H, c.scale = 40, 70000, 4096, 1.0
    elif kind == "negative":
        c.K, c.J, c.H, c.scale = -1, -30, -64, 1.0
    return c


def arguments(args, cfg_kind, ptr_kind, size):
    call, keep = [], []
    for a in args:
        if a is _hip._CFG:
            c = irt_cfg(cfg_kind)
            keep.append(c)
            call.append(ctypes.byref(c))
        elif isinstance(a, type) and issubclass(a, ctypes._Pointer) and a._type_ is _hip.HoDinaCfg:
            c = hd_cfg(cfg_kind)
            keep.append(c)
            call.append(ctypes.byref(c))
        elif isinstance(a, type) and issubclass(a, ctypes._Pointer):
            call.append(None)                         # segment tables, Adam tails: always null here
        elif a is ctypes.c_void_p:
            call.append(None if ptr_kind == "null" else ctypes.cast(HOST, ctypes.c_void_p))
        elif a is ctypes.c_float:
            call.append(1.0)
        else:
            call.append(size)
    return call, keep


def main():
    L = _hip.lib()
    bad = 0
    cases = [("zero", "null", 0), ("plausible", "null", 4096), ("too_big", "host", 4096), ("negative", "host", -7),
             ("plausible", "host", -1), ("plausible", "null", 1 << 40)]
    for name, (rt, args) in sorted(_hip.SIGNATURES.items()):
        if name in SKIP:
            continue
        for cfg_kind, ptr_kind, size in cases:
            has_cfg = any(a is _hip._CFG or (isinstance(a, type) and issubclass(a, ctypes._Pointer) and a._type_ is _hip.HoDinaCfg)
                          for a in args)
            has_ptr = any(a is ctypes.c_void_p for a in args)
            if ptr_kind == "host" and cfg_kind == "plausible" and size < 0 and not any(
                    a in (ctypes.c_int64, ctypes.c_int32) for a in args):
                continue
            if cfg_kind in ("too_big", "negative") and not has_cfg and size > 0:
                continue                              # nothing about the call is wrong then
            if ptr_kind == "null" and not has_ptr and not has_cfg:
                continue
            call, keep = arguments(args, cfg_kind, ptr_kind, size)
            sys.stdout.write("%s(%s, %s, %d) -> " % (name, cfg_kind, ptr_kind, size))
            sys.stdout.flush()
            r = getattr(L, name)(*call)
            ok = r == -1
            # a size query may answer 0 or a positive count for a plausible shape: it takes no pointer
            if not ok and not has_ptr and cfg_kind == "plausible" and r >= 0:
                ok = True
            print(r if ok else "%d  <-- not VX_EINVAL" % r)
            bad += 0 if ok else 1
    print("abi_fuzz: %d calls that were not refused" % bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
