cd /root/repo
python - <<'PY'
import os, sys, time, subprocess
sys.path.insert(0, 'tools'); sys.path.insert(0, '.')
import numpy as np, synth
from oracle import oracle as O
d = '/tmp/cpu_probe'; os.makedirs(d, exist_ok=True)
g = synth.make_genomes(64, 1000000)
synth.write_reference(d, g)
O.ref_build(d, threads=32)
names, seqs = synth.sample_reads(g, 400000, 100)
synth.write_fasta(os.path.join(d, 'reads.fa'), names, seqs)
for p in (8, 16, 32, 64, 128, 256):
    t0 = time.time()
    r = subprocess.run([os.path.join(O.REF_DIR, 'centrifuge-class'), '-f', '-t', '-p', str(p), '--reorder', '-x', os.path.join(d, 'idx'),
                    '-U', os.path.join(d, 'reads.fa'), '-S', os.path.join(d, 'o.tsv'), '--report-file', os.path.join(d, 'r.tsv')],
                   capture_output=True, text=True)
    dt = time.time() - t0
    ms = [l for l in (r.stdout + r.stderr).splitlines() if 'Multiseed' in l or 'loading' in l.lower()]
    print('p=%d wall %.2fs -> %.0f reads/s | %s' % (p, dt, 400000 / dt, ' ; '.join(ms)), flush=True)
PY
nproc; lscpu | grep -i "model name\|socket\|numa node(s)"
