"""Shape of the compiled multi-RHS matrix-core kernels (gfx950 code object inside libhmx.so), checked without a GPU.

These kernels are bound by how their stream loads are issued.  Twice in round 3 a source change that looked harmless made
the compiler move operand loads under exec-mask branches with an `s_waitcnt vmcnt(0)` behind each of them -- same registers, same LDS, 45 %
more time on the fp32 expand stage of config 5.  The counts below are what the kernels have when the loads are issued back to back."""
import collections
import os
import re
import struct
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "htool_amd", "libhmx.so")
LLVM = "/opt/rocm/lib/llvm/bin"


def code_objects(tmp_path):
    """Every gfx950 code object in libhmx.so: one offload bundle per translation unit (the engine is compiled once per coefficient type)."""
    data = open(LIB, "rb").read()
    out, i = [], data.find(b"__CLANG_OFFLOAD_BUNDLE__")
    assert i >= 0, "no offload bundle in libhmx.so"
    while i >= 0:
        off = i + 24
        n = struct.unpack_from("<Q", data, off)[0]
        off += 8
        for _ in range(n):
            o, s, l = struct.unpack_from("<QQQ", data, off)
            off += 24
            name = data[off:off + l].decode()
            off += l
            if "gfx950" in name and s > 0:
                p = tmp_path / ("hmx_gfx950_%d.co" % len(out))
                p.write_bytes(data[i + o:i + o + s])
                out.append(str(p))
        i = data.find(b"__CLANG_OFFLOAD_BUNDLE__", i + 24)
    assert out, "no gfx950 code object in libhmx.so"
    return out


def kernel_symbols(cos):
    """kernel name -> code object that defines it"""
    where = {}
    for co in cos:
        syms = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "-sW", co], capture_output=True, text=True, check=True).stdout
        for name in re.findall(r"FUNC\s+\S+\s+\S+\s+\d+\s+(\S+)$", syms, flags=re.M):
            where[name] = co
    return where


@pytest.mark.skipif(not os.path.exists(os.path.join(LLVM, "llvm-objdump")), reason="llvm-objdump not installed")
def test_staged_matrix_core_kernels_issue_their_loads_back_to_back(tmp_path):
    where = kernel_symbols(code_objects(tmp_path))
    names = sorted(n for n in where if re.fullmatch(r"\S*3(?:f64|f32)21(?:expand|reduce)_mfma(?:16|32)s_kernelILi4E\S*|\S*3(?:z64|c32)2[12](?:expand|reduce)_zmfma(?:8|16)s_kernelILi4E\S*", n))
    assert len(names) == 16, names  # expand + reduce: 16- and 32-wide real (f64, f32), 8- and 16-wide complex (z64, c32); 4 waves per workgroup (the default)
    for sym in names:
        asm = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--disassemble-symbols=" + sym, where[sym]], capture_output=True, text=True, check=True).stdout
        ops = collections.Counter(l.split()[0] for l in asm.split("\n") if l.startswith("\t"))
        drained = sum(1 for l in asm.split("\n") if "s_waitcnt vmcnt(0)" in l)
        assert ops["v_mfma_f64_16x16x4_f64"] + ops["v_mfma_f32_16x16x4_f32"] >= 32, sym
        assert ops["scratch_load_dword"] + ops["scratch_store_dword"] == 0, sym
        if re.search(r"3(?:f64|f32)21expand_mfma16s_kernel", sym):
            # round 5: every prefetch unconditional, load groups pinned in program order -- the compiler can count what is outstanding, so a step
            # is applied while the whole next step (indices, 4 gathers, 16 stream columns) stays in flight: the loop's waits are vmcnt(20 ... 36),
            # the only vmcnt(0) left are the prologue's index load and the write-out.  (Behind `if (more columns) prefetch;` every use waited for
            # the prefetch itself: one drain per 16 columns.)
            depths = [int(x) for x in re.findall(r"s_waitcnt vmcnt\((\d+)\)", asm)]
            assert drained <= 3 and max(depths) >= 30 and sum(1 for d in depths if d >= 20) >= 12, (sym, drained, sorted(depths))
        if "expand" in sym:
            # the loop has no divergent branch at all: the three are the tile loop's tail and the write-out of the rows
            assert ops["s_cbranch_execz"] <= (6 if ("mfma32s" in sym or "zmfma16s" in sym) else 4) and drained <= 8, (sym, ops["s_cbranch_execz"], drained)  # 32-wide: two write-outs
        else:
            # reduce stage: the 32 (64 for complex double) predicated stores of the write-out are branches, but the loop waits for
            # nothing but its own tile, and the destinations are fetched before the stores
            assert drained <= 12, (sym, drained)


@pytest.mark.skipif(not os.path.exists(os.path.join(LLVM, "llvm-objdump")), reason="llvm-objdump not installed")
def test_single_vector_kernels_keep_their_occupancy_and_load_shape(tmp_path):
    """The headline kernels: registers (eight waves per SIMD for the expand stage, at least seven for the reduce stage), no scratch, the
    stream loads of the main loops as wide as they were tuned to be (fp64: 16-byte loads in the reduce stage, one 8-byte column entry per
    lane in the expand stage)."""
    cos = code_objects(tmp_path)
    where = kernel_symbols(cos)
    meta = {}
    for co in cos:
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True, check=True).stdout
        for blk in notes.split("- .agpr_count:")[1:]:
            name = re.search(r"\.name:\s+(\S+)", blk).group(1)
            meta[name] = {k: int(re.search(r"\." + k + r":\s+(\d+)", blk).group(1)) for k in ("vgpr_count", "private_segment_fixed_size", "group_segment_fixed_size")}
    checked = 0
    occupancy_checked = []
    for name, m in meta.items():
        if re.search(r"3(f64|f32|z64|c32)13expand_kernelILi\d", name):
            assert m["private_segment_fixed_size"] == 0 and m["vgpr_count"] <= 64, (name, m)  # 8 waves per SIMD
            checked += 1
        if re.search(r"3(f64|f32)13reduce_kernelILi\d", name):
            assert m["private_segment_fixed_size"] == 0 and m["vgpr_count"] <= 72, (name, m)  # 7 waves per SIMD
            checked += 1
        if re.search(r"3(f64|z64)(17expand_sym_kernel|13rowsym_kernel)", name):
            # (complex double: a lane-dependent `c ? a : b` between two complex values used to become an indexed private array -- 48 bytes of
            # scratch in the hot loop, Hermitian N = 1e6 product 14.4 ms; with component-wise selects 12.7 ms)
            assert m["private_segment_fixed_size"] == 0, (name, m)
            checked += 1
        if re.search(r"aca_cb_(row|col)_kernel|sym_mfma16_kernel|rowsym_mfma16_kernel", name):
            assert m["private_segment_fixed_size"] == 0, (name, m)
        # Round 6: the stored-triangle matrix-core kernels live on their waves per SIMD.  vgpr_count is the whole allocation in the unified register
        # file (accumulation registers included): <= 256 = two waves, <= 168 = three.  The loop over a group's row ranges once cost the fp64 E pass
        # 24 registers -- 264 -- and with them a third of its speed, with identical instruction counts (profiles/r6_pmc_ab_loop.log).
        if re.search(r"3f64(24expand_sym_mfma16_kernelILi4ELb1E|20rowsym_mfma16_kernel)", name):
            assert m["vgpr_count"] <= 256, (name, m)
            occupancy_checked.append(name)
        if re.search(r"3f3224expand_sym_mfma16_kernelILi4ELb1E", name):
            assert m["vgpr_count"] <= 168, (name, m)
            occupancy_checked.append(name)
        if re.search(r"3f6421reduce_mfma16s_kernel", name):
            assert m["vgpr_count"] <= 168, (name, m)
            occupancy_checked.append(name)
        # ... and on the LDS that goes with it: tiles + a group's accumulators of two (fp32: three) workgroups per CU in 160 KB
        if re.search(r"3f6424expand_sym_mfma16_kernelILi4ELb1E", name):
            assert m["group_segment_fixed_size"] + 320 * 16 * 8 <= 80 * 1024, (name, m)
        if re.search(r"3f3224expand_sym_mfma16_kernelILi4ELb1E", name):
            assert 3 * (m["group_segment_fixed_size"] + 512 * 16 * 4) <= 160 * 1024, (name, m)
    assert len(occupancy_checked) >= 4, occupancy_checked
    assert checked >= 18, checked  # expand_kernel<4|8> x 4 types, reduce_kernel<1|4> x 2, the fused symmetric kernels (real and complex double)
    for sym, op, least in (("_ZN3hmx3f6413reduce_kernelILi1EEEvNS0_10ReduceArgsE", "global_load_dwordx4", 8), ("_ZN3hmx3f6413expand_kernelILi4EEEvNS0_10ExpandArgsE", "global_load_dwordx2", 8)):
        asm = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--disassemble-symbols=" + sym, where[sym]], capture_output=True, text=True, check=True).stdout
        ops = collections.Counter(l.split()[0] for l in asm.split("\n") if l.startswith("\t"))
        assert ops[op] >= least, (sym, op, ops[op])
