"""Test infrastructure: run the engine's HOST code (vt_engine.hip, engine.py, parallel.py) without a GPU against a recording
stand-in for the kernel library and the HIP runtime (tests/c/engine_ledger.hip), and check the recorded schedule for races.

`build()` links the product objects vt_engine.o + vt_api.o with the stand-in into tests/c/_build/libvt_engine_ledger.so.
`Ledger` loads it, exposes the log, and `races()` runs a vector-clock happens-before analysis over it:

  * every operation ticks the clock of the stream it was enqueued on;
  * an event record snapshots that stream's clock, a stream wait merges the snapshot the event held WHEN THE WAIT WAS ENQUEUED
    (hipStreamWaitEvent semantics) into the waiting stream's clock;
  * operation a happens-before a later-enqueued b iff b's clock has reached a's tick on a's stream;
  * two operations on different streams that touch overlapping bytes, at least one writing, with neither ordered before the
    other, are reported.

Nothing is executed on a device and no pointer is dereferenced: this checks the ORDER the host establishes, which is exactly
what a stream race is about."""
import ctypes
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "video-tokenizer_amd")
BUILD = os.path.join(ROOT, "tests", "c", "_build")
LIB = os.path.join(BUILD, "libvt_engine_ledger.so")
SRC = os.path.join(ROOT, "tests", "c", "engine_ledger.hip")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

MAIN, SIDE, COMM = 0x1000, 0x2000, 0x3000       # stream handles (any distinct integers: nothing dereferences them)
STREAM_NAMES = {MAIN: "main", SIDE: "side", COMM: "comm", 0: "null"}


def build():
    import importlib.util
    spec = importlib.util.spec_from_file_location("vt_build", os.path.join(PKG, "build.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    b.build()                                    # the product objects (cached)
    objs = [os.path.join(PKG, "_obj", "vt_engine.o"), os.path.join(PKG, "_obj", "vt_api.o")]
    os.makedirs(BUILD, exist_ok=True)
    stub = os.path.join(BUILD, "engine_ledger.o")
    deps = [SRC, os.path.join(PKG, "csrc", "vt_common.h"), os.path.join(ROOT, "include", "vt_hip.h")]
    if not os.path.exists(stub) or any(os.path.getmtime(d) > os.path.getmtime(stub) for d in deps):
        subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O1", "-std=c++17", "-fPIC", "-c", SRC, "-o", stub])
    if not os.path.exists(LIB) or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs + [stub]):
        # -Bsymbolic: the engine's references to hip* / vt_* bind to the definitions inside this library, whatever else the process has loaded
        subprocess.check_call([HIPCC, "-shared", "-fPIC", "-Wl,-Bsymbolic", "-o", LIB] + objs + [stub])
    return LIB


class Op:
    __slots__ = ("i", "kind", "name", "stream", "event", "ranges", "clock")

    def __repr__(self):
        return f"#{self.i} {self.name}@{STREAM_NAMES.get(self.stream, hex(self.stream))}"


class Ledger:
    def __init__(self):
        self.lib = ctypes.CDLL(build())
        L = self.lib
        L.vt_ledger_size.restype = ctypes.c_int64
        L.vt_ledger_get.restype = ctypes.c_int64
        L.vt_ledger_get.argtypes = [ctypes.c_int64, ctypes.c_char_p, ctypes.c_int64]
        L.vt_ledger_note_access.argtypes = [ctypes.c_char_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_int]
        L.vt_ledger_note_record.restype = ctypes.c_uint64
        L.vt_ledger_note_record.argtypes = [ctypes.c_uint64]
        L.vt_ledger_note_wait.argtypes = [ctypes.c_uint64, ctypes.c_uint64]

    def reset(self):
        self.lib.vt_ledger_reset()

    def ops(self):
        out = []
        buf = ctypes.create_string_buffer(1 << 16)
        for i in range(self.lib.vt_ledger_size()):
            n = self.lib.vt_ledger_get(i, buf, len(buf))
            if n < -1:
                buf = ctypes.create_string_buffer(-n + 16)
                n = self.lib.vt_ledger_get(i, buf, len(buf))
            kind, name, stream, event, rng = buf.value.decode().split("\t")
            o = Op()
            o.i, o.kind, o.name, o.stream, o.event = i, kind, name, int(stream), int(event)
            o.ranges = [tuple(int(v) for v in r.split(":")) for r in rng.split(",")] if rng else []
            out.append(o)
        return out

    # ---- what the Python side of a step adds to the same log
    def access(self, name, stream, lo, hi, write):
        self.lib.vt_ledger_note_access(name.encode(), stream, lo, hi, int(write))

    def record(self, stream):
        return self.lib.vt_ledger_note_record(stream)

    def wait(self, stream, event):
        self.lib.vt_ledger_note_wait(stream, event)


def clocks(ops):
    """Vector clocks in enqueue order.  Returns the list of problems found on the way (waits on events never recorded)."""
    vc, ev, problems = {}, {}, []
    for o in ops:
        c = vc.setdefault(o.stream, {})
        c[o.stream] = c.get(o.stream, 0) + 1
        if o.kind == "W":
            snap = ev.get(o.event)
            if snap is None:
                problems.append(f"{o!r}: waits for event {o.event} that was never recorded")
            else:
                for s, t in snap.items():
                    if c.get(s, 0) < t:
                        c[s] = t
        elif o.kind == "R":
            ev[o.event] = dict(c)
        o.clock = dict(c)
    return problems


def happens_before(a, b):
    """a was enqueued before b"""
    return a.stream == b.stream or b.clock.get(a.stream, 0) >= a.clock[a.stream]


def races(ops, limit=20):
    """Unordered conflicting accesses (see the module docstring) as readable strings, at most `limit`."""
    problems = clocks(ops)
    acc = []                                         # (lo, hi, write, op)
    for o in ops:
        if o.kind == "K":
            if o.name.startswith("UNMODELLED"):
                problems.append(f"{o!r}: a launch without a footprint in tests/c/engine_ledger.hip")
            for lo, hi, w in o.ranges:
                acc.append((lo, hi, w, o))
    acc.sort(key=lambda a: a[0])
    active, seen = [], set()
    for lo, hi, w, o in acc:
        active = [a for a in active if a[1] > lo]
        for alo, ahi, aw, ao in active:
            if ao.stream == o.stream or not (w or aw) or ao is o:
                continue
            first, second = (ao, o) if ao.i < o.i else (o, ao)
            if happens_before(first, second):
                continue
            key = (first.i, second.i)
            if key in seen:
                continue
            seen.add(key)
            problems.append(f"RACE {first!r} {'W' if (aw if first is ao else w) else 'R'} / {second!r} {'W' if (w if second is o else aw) else 'R'}"
                            f" on [{max(lo, alo):#x}, {min(hi, ahi):#x})")
        active.append((lo, hi, w, o))
    return problems[:limit], len(problems)
