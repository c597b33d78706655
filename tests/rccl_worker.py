"""One rank of tests/test_gpu_rccl2.py: the library's communicator (include/ngsqc.h ngsqc_comm_*, RCCL over xGMI) with one process per GPU.
usage: rccl_worker.py <rank> <world> <bam> <uid file> <result .npz>"""
import importlib
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
for p in (os.path.dirname(HERE), HERE):
    sys.path.insert(0, p)
ngsqc = importlib.import_module("ngs-bits_amd")
import hostprep as H  # noqa: E402


def main():
    rank, world, bam, uid_file, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5]
    if rank == 0:
        uid = ngsqc.Comm.unique_id()
        with open(uid_file + ".tmp", "wb") as f:
            f.write(uid)
        os.rename(uid_file + ".tmp", uid_file)
    else:
        t0 = time.time()
        while not os.path.exists(uid_file):
            if time.time() - t0 > 120:
                raise TimeoutError("rank 0 never published the unique id")
            time.sleep(0.05)
        uid = open(uid_file, "rb").read()
    comm = ngsqc.Comm(rank, world, uid, device=rank)
    res = {}
    try:
        assert (comm.rank, comm.world) == (rank, world)
        # the plain collectives: every rank brings (rank + 1) x a pattern
        v = (np.arange(64, dtype=np.int64) * 7 - 100) * (rank + 1)
        res["sum_i64"] = comm.allreduce_i64(v); res["max_i64"] = comm.allreduce_i64(v, take_max=True)
        res["sum_f64"] = comm.allreduce_f64(np.linspace(0.0, 1.0, 101) * (rank + 1))
        res["gathered"] = comm.allgather_summaries(np.arange(6, dtype=np.int64) + 10 * rank)
        # the shard protocol: this rank's shard of the BAM through the whole fused job, every exchange over the communicator
        h = ngsqc.Handle(path=bam, device=rank, shard=(rank, world))
        omim = os.path.join(os.path.dirname(HERE), "ngs-bits_amd", "resources", "hg38_440_omim_genes.bed")
        regs, _ = H.bed_regions(omim, h.refs, 3); tx, ty = H.xy_tids(h.refs)
        kw = dict(regions=regs, min_mapq=1, tid_x=tx, tid_y=ty, nonspecial=H.nonspecial(h.refs))
        counters, gc, summaries, site_counts = ngsqc.scan_mapping_sharded(h, ngsqc.MODE_WGS, comm=comm, sites=H.known_sites(h.refs), **kw)
        res["counters"] = np.asarray(counters); res["gc"] = np.asarray(gc); res["summaries"] = np.asarray(summaries); res["site_counts"] = np.asarray(site_counts)
        res["depth"] = h.depth(int(counters[26])).copy()
        h.close()
    finally:
        comm.close()
    np.savez(out, **res)


if __name__ == "__main__":
    main()
