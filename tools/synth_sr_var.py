#!/usr/bin/env python3
"""short reads of mixed lengths (60-600 bp) for the ShortReads variant: both geometry branches (qlen <= 300 / > 300), reads
flush with or hanging near the contig ends, exact copies, Ns.   synth_sr_var.py SEED N out.fq [ref.fa]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import synth  # noqa: E402

names, contigs = synth.read_fasta(sys.argv[4] if len(sys.argv) > 4 else "ref.fa")
rng = np.random.default_rng(int(sys.argv[1]))
out=[]
for i in range(int(sys.argv[2])):
    c = int(rng.integers(0,len(contigs)))
    ln = int(rng.choice([60,100,150,151,250,299,300,301,400,600]))
    mode = i % 10
    L = len(contigs[c])
    if mode == 0: st = 0
    elif mode == 1: st = L - ln
    elif mode == 2: st = int(rng.integers(0, 30))
    elif mode == 3: st = L - ln - int(rng.integers(0,30))
    else: st = int(rng.integers(0, L-ln))
    s = synth.mutate(rng, contigs[c][st:st+ln], 0.02 if i%3 else 0.0, 0.002, 0.002)
    if i % 17 == 0 and len(s) > 20: s = s.copy(); s[int(rng.integers(0,len(s)))] = 78
    rev = rng.random()<0.5
    if rev: s = np.array([synth.COMP[x] for x in s[::-1]],dtype=np.uint8)
    out.append(("v_%d_c%d_%d_%s"%(i,c+1,st,"-" if rev else "+"), s))
synth.write_fastq(sys.argv[3], out)
