"""Runs tools/probe.so (see tools/probe.hip): prints what LDS-DMA wrote for in-range and out-of-range source offsets."""
import ctypes, torch, os
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "probe.so"))
src = torch.arange(1, 257, dtype=torch.int32, device="cuda")
out = torch.zeros(1024, dtype=torch.int32, device="cuda")
lib.run_probe.argtypes = [ctypes.c_void_p, ctypes.c_uint, ctypes.c_void_p]
print("rc", lib.run_probe(src.data_ptr(), src.numel() * 4, out.data_ptr()))
o = out.cpu().view(torch.int32)
print("first instr, in-range lanes  :", o[:8].tolist(), "...", o[124:128].tolist())
print("first instr, OOB lanes (32..):", [hex(v & 0xffffffff) for v in o[128:136].tolist()], [hex(v & 0xffffffff) for v in o[252:256].tolist()])
print("second instr in-range        :", o[256:264].tolist())
print("second instr OOB             :", [hex(v & 0xffffffff) for v in o[384:392].tolist()])
print("untouched                    :", [hex(v & 0xffffffff) for v in o[512:516].tolist()])
