"""CPU-side checks (no GPU): the C-ABI library loads and exports every symbol of include/ngsqc.h, fails loudly without a
device, the host tools parse their CLI like the reference, and the multi-process counter reduction works on gloo."""
import ctypes as C
import importlib
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLDEN_IN as GI, ROOT

ngsqc = importlib.import_module("ngs-bits_amd")
BIN = os.path.join(ROOT, "ngs-bits_amd", "bin")


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "ngsqc.h")).read()
    declared = sorted(set(re.findall(r"\b(ngsqc_[a-z_0-9]+)\s*\(", hdr)))
    assert len(declared) >= 20
    L = ngsqc.lib()
    missing = [s for s in declared if not hasattr(L, s)]
    assert not missing, missing
    assert b"gfx950" in L.ngsqc_version()
    L.ngsqc_abi_version.restype = C.c_int32
    assert L.ngsqc_abi_version() == 6 and b"abi 6" in L.ngsqc_version()   # (the round of include/ngsqc.h the library was built from: ngsqc_timings grows at its end)
    # ngsqc_device_count never fails: 0 on a box without a device (this container), the library's own count elsewhere - what tests/test_gpu_rccl2.py arms itself by
    n_dev = ngsqc.device_count()
    assert n_dev >= 0 and (n_dev == 0 or os.path.exists("/dev/kfd"))


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(ngsqc.NgsqcError) as e:
        ngsqc.Handle(path=os.path.join(GI, "close_exons.bam"))
    assert e.value.code == -4 and "no CPU fallback" in str(e.value)


def test_communicator_needs_a_device():
    """ngsqc_comm_init without a HIP device: the same loud error as every other entry point (the collective has no CPU path either)"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(ngsqc.NgsqcError) as e:
        ngsqc.Comm(0, 1, bytes(128), device=0)
    assert e.value.code == -4 and "no CPU fallback" in str(e.value)


def test_special_parameters_of_the_tools(tmp_path):
    """ToolBase's special parameters (doc/tools/MappingQC.md:40-45): --changelog lists the reference's entries, --tdx writes <tool>.tdx, --settings names the only settings file"""
    r = _tool("MappingQC", "--changelog")
    assert r.returncode == 0 and "2023-05-12 Added 'read_qc' parameter." in r.stdout and r.stdout.startswith("MappingQC ")
    r = subprocess.run([os.path.join(BIN, "BedCoverage"), "--tdx"], capture_output=True, text=True, cwd=str(tmp_path), timeout=60)
    tdx = (tmp_path / "BedCoverage.tdx").read_text()
    assert r.returncode == 0 and '<Tool name="BedCoverage"' in tdx and '<Infile name="bam">' in tdx or "InfileList" in tdx
    r = _tool("MappingQC", "--help")
    assert "--changelog" in r.stdout and "--tdx" in r.stdout and "--settings [file]" in r.stdout
    r = _tool("MappingQC", "-in", os.path.join(GI, "close_exons.bam"), "-wgs", "--settings", str(tmp_path / "missing.ini"))
    assert r.returncode == 1 and "does not exist" in r.stderr
    ini = tmp_path / "s.ini"; ini.write_text("reference_genome = /nowhere/genome.fa\n")
    r = _tool("MappingQC", "-in", os.path.join(GI, "close_exons.bam"), "-wgs", "--settings", str(ini))
    assert r.returncode == 1 and "Command line parsing exception" not in r.stderr   # (parsed; it fails later: no device here, or the genome file)


def _tool(name, *args):
    exe = os.path.join(BIN, name)
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "ngs-bits_amd", "host"), "-s"])
    return subprocess.run([exe] + list(args), capture_output=True, text=True, timeout=300)


def test_tool_cli_contract():
    bam = os.path.join(GI, "close_exons.bam")
    p = _tool("MappingQC", "--help")
    assert p.returncode == 0
    for flag in ("-in", "-out", "-roi", "-wgs", "-rna", "-txt", "-min_mapq", "-no_cont", "-debug", "-build", "-ref", "-cfdna",
                 "-somatic_custom_bed", "-read_qc", "-single_end"):  # src/MappingQC/main.cpp:23-39
        assert f"  {flag}" in p.stdout, flag
    p = _tool("MappingQC", "-in", bam, "-wgs", "-rna", "-no_ref")
    assert p.returncode != 0 and "You have to use exactly one of the parameters 'roi', 'wgs', or 'rna' !" in p.stderr
    p = _tool("MappingQC", "-in", bam, "-wgs", "-cfdna", "-no_ref")
    assert p.returncode != 0 and "The flag 'cfdna' can only be used with parameter 'roi'!" in p.stderr
    p = _tool("MappingQC", "-wgs")
    assert p.returncode != 0 and "Mandatory parameter 'in' not given." in p.stderr
    p = _tool("MappingQC", "-in", bam, "-wgs", "-build", "hg20")
    assert p.returncode != 0 and "is not valid" in p.stderr
    p = _tool("BedCoverage", "--help")
    for flag in ("-bam", "-min_mapq", "-in", "-decimals", "-out", "-ref", "-clear", "-threads", "-random_access", "-debug", "-skip_mismapped"):
        assert f"  {flag}" in p.stdout, flag   # src/BedCoverage/main.cpp:19-31
    for tool in ("BedLowCoverage", "BedHighCoverage"):
        p = _tool(tool, "--help")
        for flag in ("-bam", "-cutoff", "-in", "-random_access", "-out", "-min_mapq", "-min_baseq", "-ref", "-threads", "-debug"):
            assert f"  {flag}" in p.stdout, (tool, flag)
        p = _tool(tool, "-bam", bam)
        assert p.returncode != 0 and "Mandatory parameter 'cutoff' not given." in p.stderr


def test_sibling_tool_cli_contract_and_no_device_error():
    bam = os.path.join(GI, "close_exons.bam")
    p = _tool("SampleGender", "--help")
    for flag in ("-in", "-out", "-method", "-max_female", "-min_male", "-min_female", "-max_male", "-sry_cov", "-build", "-ref", "-long_read"):
        assert f"  {flag}" in p.stdout, flag   # src/SampleGender/main.cpp:22-36
    p = _tool("SampleGender", "-in", bam, "-method", "zz")
    assert p.returncode != 0 and "is not valid" in p.stderr
    p = _tool("SampleGender", "-in", bam)
    assert p.returncode != 0 and "Mandatory parameter 'method' not given." in p.stderr
    p = _tool("BedReadCount", "--help")
    for flag in ("-bam", "-min_mapq", "-in", "-out", "-ref"):
        assert f"  {flag}" in p.stdout, flag   # src/BedReadCount/main.cpp:23-30
    import torch
    if not torch.cuda.is_available():   # no CPU fallback in any tool
        for args in (("SampleGender", "-in", bam, "-method", "xy"), ("BedReadCount", "-bam", bam, "-in", os.path.join(GI, "close_exons.bed"))):
            p = _tool(*args)
            assert p.returncode != 0 and "no CPU fallback" in p.stderr, p.stderr


def test_bench_spawns_its_ranks(tmp_path):
    """`python bench.py --gpus N` without a launcher must start N ranks itself (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*): checked with a stub that
    replaces main() by printing the rank environment (no GPU needed)."""
    stub = tmp_path / "stub.py"
    stub.write_text(
        "import importlib.util, os, sys\n"
        f"spec = importlib.util.spec_from_file_location('bench', {os.path.join(ROOT, 'bench.py')!r}); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)\n"
        "if 'WORLD_SIZE' in os.environ:\n"
        "    print('rank', os.environ['RANK'], os.environ['LOCAL_RANK'], os.environ['WORLD_SIZE'], os.environ['MASTER_ADDR'], flush=True); sys.exit(0)\n"
        "b.__file__ = __file__\n"
        "sys.exit(b.spawn_ranks(3))\n")
    p = subprocess.run([sys.executable, str(stub), "--gpus", "3"], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr
    assert p.stdout.split() == ["rank", "0", "0", "3", "127.0.0.1"]   # rank 0's stdout passes through, the other ranks are silenced


def test_combine_counters_semantics():
    rng = np.random.default_rng(0)
    a, b = rng.integers(0, 1000, ngsqc.NCOUNTERS), rng.integers(0, 1000, ngsqc.NCOUNTERS)
    c = ngsqc.combine_counters_local([a, b])
    assert c[0] == a[0] + b[0] and c[24] == max(a[24], b[24]) and c[25] == max(a[25], b[25]) and c[1000] == a[1000] + b[1000]
    assert [ngsqc.shard_blocks(10, 4, r) for r in range(4)] == [(0, 3), (3, 6), (6, 8), (8, 10)]


_WORKER = r"""
import importlib, os, sys
import numpy as np
import torch.distributed as dist
sys.path.insert(0, sys.argv[1])
ngsqc = importlib.import_module("ngs-bits_amd")
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group(backend="gloo", rank=rank, world_size=world)
vecs = [np.random.default_rng(100 + r).integers(0, 10**9, ngsqc.NCOUNTERS) for r in range(world)]
got = ngsqc.allreduce_counters(vecs[rank])
exp = ngsqc.combine_counters_local(vecs)
assert np.array_equal(got, exp), "rank %d mismatch" % rank
dist.barrier()
dist.destroy_process_group()
print("ok", rank)
"""


def test_counter_allreduce_gloo_world2(tmp_path):
    """The N>1 path of bench.py: one process per rank, one collective over the counter vectors (gloo here, RCCL on GPUs)."""
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29531", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    for p in procs:
        out, err = p.communicate(timeout=240)
        assert p.returncode == 0, err[-2000:]
        assert "ok" in out


@pytest.mark.parametrize("mode", [0, 1, 2, 3, 4, 5])
def test_region_tables_of_the_host_layer_equal_the_oracle(mode):
    """bench.py and the C-ABI tests take their region tables from the product's BedFile / Chromosome code (bin/libngsqc_hostapi.so); the oracle's BED loader is the checker:
    every fixture BED x every operation the reference applies to a ROI (as loaded, merge, merge with names, sort + merge, chunk(100) of both)"""
    import glob
    import hostprep as H
    refs = [("chr%s" % c, 1) for c in list(range(1, 23)) + ["X", "Y", "M"]] + [("GL000192.1", 1)]
    beds = sorted(glob.glob(os.path.join(GI, "*.bed"))) + [os.path.join(ROOT, "ngs-bits_amd", "resources", "hg38_440_omim_genes.bed")]
    assert len(beds) >= 5
    n = 0
    for bed in beds:
        try:
            exp, _ = H.bed_regions_oracle(bed, refs, mode)
        except Exception:
            continue   # (a fixture the reference's loader refuses)
        got, _ = H.bed_regions(bed, refs, mode)
        assert got == exp, (bed, mode)
        n += len(got)
    assert n > 1000


def test_bed_positions_parse_like_qbytearray_toint(tmp_path):
    """BedFile::load (src/cppNGS/BedFile.cpp:155-160) converts with QByteArray::toInt(&ok): base 10, white space around the number is ignored, a value outside the range of
    int is NOT ok - the line is refused ("BED file line with invalid ... position found"), never wrapped"""
    import hostprep as H
    refs = [("chr1", 1), ("chr2", 1)]
    p = str(tmp_path / "a.bed")
    open(p, "w").write("chr1\t 100\t200 \nchr2\t+5\t7\r\n")
    assert H.bed_regions(p, refs, 0)[0] == [(0, 101, 200), (1, 6, 7)]
    for bad, what in (("chr1\t2147483648\t5\n", "starts"), ("chr1\t5\t-2147483649\n", "end"), ("chr1\t5\t99999999999999999999\n", "end"), ("chr1\t1e3\t5\n", "starts"),
                      ("chr1\t0x10\t50\n", "starts"), ("chr1\t\t5\n", "starts"), ("chr1\t5\t6x\n", "end")):
        open(p, "w").write("chr2\t1\t2\n" + bad)
        with pytest.raises(RuntimeError) as e:
            H.bed_regions(p, refs, 0)
        assert "invalid %s position" % what in str(e.value), (bad, str(e.value))
