"""CPU: the code objects of the last build (optable_amd/csrc/build/*.o, cross-compiled for gfx950) — no kernel uses
scratch memory or spills a vector register, the library stays small, and the kernel-argument layout that
k_trace_rolling reads its ray pointers from (kernels.h: LeadArgs) is what the code object records."""
import glob
import os
import re
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
BUILD = os.path.join(ROOT, "optable_amd", "csrc", "build")


@pytest.fixture(scope="module")
def kernels():
    if not glob.glob(os.path.join(BUILD, "*.o")):
        import __graft_entry__ as g

        g.build()
    objs = sorted(glob.glob(os.path.join(BUILD, "*.o")))
    assert objs, "no objects: run `make -C optable_amd/csrc`"
    out = []
    with tempfile.TemporaryDirectory() as tmp:
        for o in objs:
            fat, co = os.path.join(tmp, "fat.bin"), os.path.join(tmp, "dev.co")
            subprocess.check_call([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", o, fat])
            subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--type=o", f"--input={fat}", "--unbundle",
                                   "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"])
            notes = subprocess.check_output([f"{LLVM}/llvm-readelf", "--notes", co], text=True)
            # one block per kernel: "- .agpr_count ... .args: ... .name: ..." in the amdhsa.kernels list
            for block in notes.split("\n  - .agpr_count")[1:]:
                def field(name):
                    m = re.search(rf"\.{name}:\s+(\S+)", block)
                    return m.group(1) if m else None
                args = [(int(a), int(b)) for a, b in re.findall(r"\.offset:\s+(\d+)\s+\.size:\s+(\d+)", block)]
                out.append({"name": field("name"), "scratch": int(field("private_segment_fixed_size")), "vgpr": int(field("vgpr_count")),
                            "vgpr_spill": int(field("vgpr_spill_count")), "sgpr_spill": int(field("sgpr_spill_count")), "args": args})
    return out


def test_no_kernel_uses_scratch_or_spills_vector_registers(kernels):
    # One exception, bounded: the double-precision all-features tree kernel that reads its scene image from global memory (scenes no
    # LDS holds: one wave per SIMD, 64-bit addresses for every node access) parks a few values in accumulation registers — no scratch
    # memory, and the launches it serves are a handful of trees deep, not wide (fixture g27).
    def tolerated(k):
        return "k_trace_treesIdLj1023ELi1E9SegPlanesIdELi0E" in k["name"] and k["vgpr_spill"] <= 16 and not k["scratch"]

    bad = [(k["name"], k["scratch"], k["vgpr_spill"]) for k in kernels if (k["scratch"] or k["vgpr_spill"]) and not tolerated(k)]
    assert not bad, bad


def test_library_stays_small(kernels):
    # round 2 shipped 461 kernels (338 of them a scan library's tuning variants) and took two minutes to build
    assert 40 <= len(kernels) < 172, len(kernels)  # round 4: 169 (+ look-ahead emit, recount, the lane-per-tree kernels of every preset)
    names = [k["name"] for k in kernels]
    assert not any("rocprim" in n or "hipcub" in n for n in names)
    assert sum("k_trace_rolling" in n for n in names) <= 48


def test_rolling_kernel_reads_its_ray_pointers_where_the_code_object_puts_them(kernels):
    """kernels.h reads the caller's 15 ray pointers from the kernel-argument segment at offsetof(LeadArgs, in) = 56
    (SceneBlob 48 bytes, unit padded to 8): argument 2 of every k_trace_rolling instantiation must sit exactly there."""
    rolling = [k for k in kernels if "k_trace_rolling" in k["name"] or "k_trace_pool" in k["name"]]
    assert rolling
    for k in rolling:
        assert k["args"][0] == (0, 48) and k["args"][2] == (56, 120), (k["name"], k["args"][:3])
        # ... and the output descriptor (14 array pointers, or the plane block) right behind n and K
        assert k["args"][5][0] == 192 and k["args"][5][1] in (112, 16), (k["name"], k["args"][5])


def test_rolling_kernel_argument_segment_is_laid_out_like_the_struct_the_kernel_indexes(kernels):
    """k_trace_rolling loads the arguments it needs once per pass or less (queue, n, the append cursor and bounds,
    seg_count) from the argument segment at offsetof(LeadArgs, field).  LeadArgs lists the 16 parameters in order, so
    the code object must place every explicit argument where a C struct of members of those sizes places it: members
    of 8 bytes and more (pointers, int64, structs of pointers) on 8, the 4-byte ones on 4, nothing packed or padded
    otherwise."""
    rolling = [k for k in kernels if "k_trace_rolling" in k["name"] or "k_trace_pool" in k["name"]]
    assert any("k_trace_pool" in k["name"] for k in rolling)
    for k in rolling:
        explicit = k["args"][:16]
        assert len(explicit) == 16, (k["name"], len(k["args"]))
        at = 0
        for idx, (offset, size) in enumerate(explicit):
            align = 8 if size >= 8 else 4
            at = (at + align - 1) // align * align
            assert offset == at, (k["name"], idx, offset, at)
            at += size
        sizes = [s for _, s in explicit]
        assert sizes[6] == 24 and sizes[7] == 8 and sizes[13] == 8, (k["name"], sizes)  # AppendCtl, seg_count, queue


def test_everyday_presets_keep_their_register_budgets(kernels):
    """The presets that light scenes made of the reference's everyday parts select (tables.h FE = 63: polygon / boolean
    apertures, spheres, aspheres, count gates; FM = 319: + cylinders, tilted polygons, series) instead of the all-features
    instantiation: FE within 128 registers in single precision (4 waves per SIMD) and 192 in double (2), FM within 144 /
    256; the all-features lane-per-ray kernel they replace sits at 169-181 with 146-186 scalar spills."""
    def of(family, real, mask):
        return [k for k in kernels if re.match(rf"_Z\d+{family}I{real}Lj{mask}E", k["name"])]

    for family in ("k_trace_fused", "k_gen_pass", "k_gen_probe"):
        fe32, fe64 = of(family, "f", 63), of(family, "d", 63)
        fm32, fm64 = of(family, "f", 319), of(family, "d", 319)
        assert fe32 and fe64 and fm32 and fm64, family
        assert max(k["vgpr"] for k in fe32) <= 128, [(k["name"][:40], k["vgpr"]) for k in fe32]
        assert max(k["vgpr"] for k in fe64) <= 192, [(k["name"][:40], k["vgpr"]) for k in fe64]
        assert max(k["vgpr"] for k in fm32) <= 144, [(k["name"][:40], k["vgpr"]) for k in fm32]
        assert max(k["vgpr"] for k in fm64) <= 256, [(k["name"][:40], k["vgpr"]) for k in fm64]
    all32 = of("k_trace_fused", "f", 1023)
    assert all32 and min(k["vgpr"] for k in all32) > 150  # (what the split is measured against)


def test_tree_kernels_keep_the_registers_their_launch_plans_for(kernels):
    """k_trace_trees is launched with as many 256-thread workgroups per CU as tables.h tree_groups_by_registers says its registers
    allow (and the queues' LDS leaves room for): single precision 6 for the planar preset FB = 28 (80 registers), 4 for FC = 732,
    FE = 63, FM = 319 (128), 3 for the all-features preset 1023 (168); double precision 3 for FB, FC and FE (168), 2 for FM
    (256), 1 for all features."""
    def of(real, mask):
        return [k for k in kernels if re.match(rf"_Z\d+k_trace_treesI{real}Lj{mask}E", k["name"])]

    for real, mask, most in (("f", 28, 80), ("f", 732, 128), ("f", 63, 128), ("f", 319, 128), ("f", 1023, 168), ("d", 28, 168), ("d", 732, 168), ("d", 63, 168), ("d", 319, 256),
                             ("d", 1023, 512)):
        ks = of(real, mask)
        assert ks and max(k["vgpr"] for k in ks) <= most, (real, mask, [(k["name"][:40], k["vgpr"]) for k in ks])


def test_production_kernels_keep_their_scalar_spills_in_check(kernels):
    """Scalar registers that do not fit are kept in lanes of a vector register (v_writelane / v_readlane: 4-cycle vector instructions
    on gfx950).  The kernels the BASELINE configs select: cfg 2 / cfg 4 (lane per ray, double) and the lane-per-tree kernel within 32;
    the two heavy single-precision kernels carry more (the pass loop of cfg 3: 40; the block pool of cfg 5: 55 — none of them inside
    the search) and must not grow."""
    def spills(pattern):
        ks = [k for k in kernels if re.match(pattern, k["name"])]
        assert ks, pattern
        return max(k["sgpr_spill"] for k in ks)

    assert spills(r"_Z13k_trace_fusedIdLj24ELb1ELi4ELb1E") <= 32          # cfg 2: FA, image in LDS, 4 waves per SIMD, non-temporal
    assert spills(r"_Z13k_trace_fusedIdLj28ELb1ELi1ELb1E") <= 32          # cfg 4: FB
    assert spills(r"_Z13k_trace_treesIdLj28ELi\dE9SegPlanes") <= 32       # cfg 4 with reflectivity: ray trees
    assert spills(r"_Z15k_trace_rollingIfLj1180ELb1ELb1ELb1E9SegPlanes") <= 44   # cfg 3
    assert spills(r"_Z12k_trace_poolIfLj86ELb1E9SegPlanes") <= 60                # cfg 5
