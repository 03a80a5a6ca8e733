"""Native ingest throughput on this host: N copies of the serine peptidase PDB fixtures parsed with T threads."""
import os, shutil, sys, time, glob
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from folddisco_amd import structure
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
src = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "serine_peptidases", "*.pdb")))
d = "/tmp/fd_ingest_bench"
shutil.rmtree(d, ignore_errors=True); os.makedirs(d)
paths = []
for k in range(N):
    p = os.path.join(d, f"{k:06d}.pdb"); shutil.copy(src[k % len(src)], p); paths.append(p)
mb = sum(os.path.getsize(p) for p in paths) / 1e6
structure.read_compact_structures(paths[:64], threads=8)
for T in (1, 16, 64, os.cpu_count() or 1):
    t = time.perf_counter(); s, ok = structure.read_compact_structures(paths, threads=T); dt = time.perf_counter() - t
    print(f"threads {T:4d}: {N / dt:9.0f} files/s  {mb / dt:8.0f} MB/s  ({sum(x.n for x in s)} residues, CompactStructure objects)")
    t = time.perf_counter(); ps, nres, *_ = structure.read_packed(paths, threads=T); dt = time.perf_counter() - t
    print(f"threads {T:4d}: {N / dt:9.0f} files/s  {mb / dt:8.0f} MB/s  ({int(nres.sum())} residues, flat batch)")
shutil.rmtree(d, ignore_errors=True)
