"""Capture one training step into a HIP graph WITHOUT replaying it and list what the graph holds: node kinds, kernel names, and
every memcpy / memset / host node with its operands and the memory type of its pointers -- a pageable host pointer in a memcpy
node is read again at every replay, long after the host buffer is gone.  Walks the hipGraph_t with the HIP runtime API (ctypes).
Usage: python tools/graph_nodes.py ARCH"""
import collections
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from noise_robust_vit_amd.train import TrainConfig, Trainer  # noqa: E402

arch = sys.argv[1] if len(sys.argv) > 1 else "mae_b_16"
batch = int(os.environ.get("batch", "64"))
dev = torch.device("cuda", 0)
kind, kw, _ = bench.ARCHS[arch]
model = bench.build_model(arch).to(dev).train()
trainer = Trainer(model, TrainConfig(lr=5e-4, weight_decay=0.05, grad_max_norm=5.0), None,
                  compute_loss=(lambda m, xb, yb: m(xb)) if kind == "mae" else None)
g = torch.Generator(device=dev).manual_seed(1234)
x = torch.randn(batch, 3, 224, 224, generator=g, device=dev).to(torch.bfloat16)
y = torch.randint(0, 1000, (batch,), generator=g, device=dev)
for _ in range(2):
    trainer.step(x, y)
torch.cuda.synchronize()
trainer.capture(x, y, keep_graph=True)
graph = trainer._graph
trainer._graph = None                   # never replayed here
torch.cuda.synchronize()

hip = C.CDLL("libamdhip64.so")
hip.hipGraphGetNodes.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
hip.hipGraphNodeGetType.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
hip.hipGraphKernelNodeGetParams.argtypes = [C.c_void_p, C.c_void_p]
hip.hipGraphMemcpyNodeGetParams.argtypes = [C.c_void_p, C.c_void_p]
hip.hipGraphMemsetNodeGetParams.argtypes = [C.c_void_p, C.c_void_p]
hip.hipPointerGetAttributes.argtypes = [C.c_void_p, C.c_void_p]
hgraph = C.c_void_p(graph.raw_cuda_graph())
n = C.c_size_t(0)
assert hip.hipGraphGetNodes(hgraph, None, C.byref(n)) == 0
nodes = (C.c_void_p * n.value)()
assert hip.hipGraphGetNodes(hgraph, nodes, C.byref(n)) == 0
TYPES = ["kernel", "memcpy", "memset", "host", "graph", "empty", "wait_event", "event_record", "sem_signal", "sem_wait",
         "mem_alloc", "mem_free", "memcpy_from_symbol", "memcpy_to_symbol", "batch_mem_op"]


class Dim3(C.Structure):
    _fields_ = [("x", C.c_uint), ("y", C.c_uint), ("z", C.c_uint)]


class KernelParams(C.Structure):
    _fields_ = [("blockDim", Dim3), ("extra", C.c_void_p), ("func", C.c_void_p), ("gridDim", Dim3),
                ("kernelParams", C.c_void_p), ("sharedMemBytes", C.c_uint)]


class Pos(C.Structure):
    _fields_ = [("x", C.c_size_t), ("y", C.c_size_t), ("z", C.c_size_t)]


class Pitched(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("pitch", C.c_size_t), ("xsize", C.c_size_t), ("ysize", C.c_size_t)]


class Memcpy3D(C.Structure):
    _fields_ = [("srcArray", C.c_void_p), ("srcPos", Pos), ("srcPtr", Pitched), ("dstArray", C.c_void_p), ("dstPos", Pos),
                ("dstPtr", Pitched), ("extent", Pos), ("kind", C.c_int)]


class MemsetParams(C.Structure):
    _fields_ = [("dst", C.c_void_p), ("elementSize", C.c_uint), ("height", C.c_size_t), ("pitch", C.c_size_t),
                ("value", C.c_uint), ("width", C.c_size_t)]


class PtrAttr(C.Structure):
    _fields_ = [("type", C.c_int), ("device", C.c_int), ("devicePointer", C.c_void_p), ("hostPointer", C.c_void_p),
                ("isManaged", C.c_int), ("allocationFlags", C.c_uint)]


def memtype(p):
    if not p:
        return "null"
    a = PtrAttr()
    rc = hip.hipPointerGetAttributes(C.byref(a), C.c_void_p(p))
    if rc != 0:
        hip.hipGetLastError()
        return f"UNREGISTERED (pageable host or freed; hipPointerGetAttributes rc {rc})"
    return {0: "unregistered", 1: "pinned host", 2: "device", 3: "managed"}.get(a.type, str(a.type))


hip.hipKernelNameRefByPtr.restype = C.c_char_p
hip.hipKernelNameRefByPtr.argtypes = [C.c_void_p, C.c_void_p]
kinds = collections.Counter()
knames = collections.Counter()
KINDS = {0: "H2H", 1: "H2D", 2: "D2H", 3: "D2D", 4: "default"}
print(f"{arch}: {n.value} nodes")
for i in range(n.value):
    t = C.c_int(-1)
    rc_t = hip.hipGraphNodeGetType(nodes[i], C.byref(t))
    if rc_t != 0 and i == 0:
        print('hipGraphNodeGetType rc', rc_t)
    name = TYPES[t.value] if 0 <= t.value < len(TYPES) else str(t.value)
    kinds[name] += 1
    if name == "kernel":
        kp = KernelParams()
        if hip.hipGraphKernelNodeGetParams(nodes[i], C.byref(kp)) == 0 and kp.func:
            s = hip.hipKernelNameRefByPtr(kp.func, None)
            knames[(s.decode() if s else "?")[:90]] += 1
        else:
            knames["<no params>"] += 1
    elif name == "memcpy":
        mp = Memcpy3D()
        rc = hip.hipGraphMemcpyNodeGetParams(nodes[i], C.byref(mp))
        print(f"  node {i}: memcpy rc={rc} kind={KINDS.get(mp.kind, mp.kind)} bytes={mp.extent.x * max(mp.extent.y, 1) * max(mp.extent.z, 1)} "
              f"src={mp.srcPtr.ptr and hex(mp.srcPtr.ptr)} [{memtype(mp.srcPtr.ptr)}] dst={mp.dstPtr.ptr and hex(mp.dstPtr.ptr)} [{memtype(mp.dstPtr.ptr)}]")
    elif name == "memset":
        ms = MemsetParams()
        rc = hip.hipGraphMemsetNodeGetParams(nodes[i], C.byref(ms))
        print(f"  node {i}: memset rc={rc} dst={ms.dst and hex(ms.dst)} [{memtype(ms.dst)}] elem={ms.elementSize} width={ms.width} height={ms.height} value={ms.value}")
    elif name != "empty":
        print(f"  node {i}: {name}")
print("node kinds:", dict(kinds))
# hipGraphMemcpyNodeGetParams returns nothing usable for the 1-D memcpy nodes a captured hipMemcpyAsync becomes: let the runtime
# print them (hipGraphDebugDotPrint), and show every node of the dot file that is not a kernel
dot = os.path.join(ROOT, "gpurun_out", f"graph_{arch}.dot")
os.makedirs(os.path.dirname(dot), exist_ok=True)
hip.hipGraphDebugDotPrint.argtypes = [C.c_void_p, C.c_char_p, C.c_uint]
rc = hip.hipGraphDebugDotPrint(hgraph, dot.encode(), 1 | (1 << 3) | (1 << 4) | (1 << 10))
print("hipGraphDebugDotPrint rc", rc, "exists", os.path.exists(dot))
if os.path.exists(dot):
    import re
    txt = open(dot).read()
    for lab in re.findall(r'label="([^"]*)"', txt):
        low = lab.lower()
        if "memcpy" in low or "memset" in low or "htod" in low or "dtoh" in low or "host" in low:
            print("  dot:", lab.replace("\\n", " | ").replace("\n", " | ")[:500])
    os.remove(dot)
for k, v in knames.most_common(80):
    print(f"  {v:5d}  {k}")
