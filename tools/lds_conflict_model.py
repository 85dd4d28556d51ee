#!/usr/bin/env python3
"""LDS-array cycles of ONE frame of the one-frame-per-wave MFCC kernel, instruction by instruction, from the kernel's own
tables -- no GPU needed.

The bank model is MI355X_MICROARCH.md's ("LDS"): per instruction kind, the fixed lane groups a wave64 access is serviced in,
the bank of a byte address, one LDS cycle per group when conflict-free, +1 per extra DISTINCT dword address on a busy bank
(identical addresses broadcast).  The per-lane addresses are the ones mfcc_wave_device.h forms (exchange layout, power-slot
skew, filterbank run starts, ...), with the run tables of the real filterbank (dumped from mfcc_wave_tables.h by a small C++
helper this script compiles).

    python tools/lds_conflict_model.py [--real f64|f32] [--xchg stride5|swizzle] [--json out.json]

Used in round 5 to find where the 18 % conflict cycles of the float64 frame (PMC: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE)
come from before touching the kernel, and to check a candidate layout on paper (profiles/round5/r5_lds_budget_*.txt).
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np

# lane groups (one LDS cycle each when conflict-free), bank count, per instruction kind
G2x32 = [list(range(0, 32)), list(range(32, 64))]
G4x16_b128 = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
G4x16_b128 = G4x16_b128 + [[l + 32 for l in g] for g in G4x16_b128]
G4x16 = [list(range(16 * i, 16 * i + 16)) for i in range(4)]
G8x8 = [list(range(8 * i, 8 * i + 8)) for i in range(8)]
KINDS = {
    # name: (groups, banks, bytes per lane, issue cycles of the instruction itself (data transfer for stores))
    'ds_read_b32': (G2x32, 32, 4, 2), 'ds_read_b64': (G2x32, 64, 8, 2), 'ds_read_b128': (G4x16_b128, 64, 16, 4),
    'ds_write_b32': (G2x32, 32, 4, 4), 'ds_write_b64': (G4x16, 32, 8, 6), 'ds_write_b128': (G8x8, 32, 16, 13),
}


def array_cycles(kind, addrs):
    """addrs: per lane byte address or None (lane inactive) -> (LDS-array cycles, conflict-free cycles)"""
    groups, banks, nbytes, _ = KINDS[kind]
    total, ideal = 0, 0
    for g in groups:
        per_bank = {}
        active = False
        for l in g:
            a = addrs[l]
            if a is None:
                continue
            active = True
            for d in range(nbytes // 4):
                dw = a // 4 + d
                per_bank.setdefault(dw % banks, set()).add(dw)
        if active:
            total += max(len(s) for s in per_bank.values())
            ideal += 1
    return total, ideal


def dump_tables(real):
    from mycroft_precise_amd.vectorization import mel_filterbank
    src = r'''
#include <cstdio>
#include <fstream>
#include "mfcc_wave_core.h"
using namespace pe_wave;
template <class R> int run(const char* ffilt) {
    std::ifstream a(ffilt, std::ios::binary);
    std::vector<double> filt((size_t)20 * kBins);
    a.read(reinterpret_cast<char*>(filt.data()), (std::streamsize)(filt.size() * 8));
    std::vector<unsigned char> blob; Layout L;
    if (!build<R>(filt.data(), 20, 13, blob, L).empty()) return 1;
    const Tab<R> t = bind<R>(blob.data(), L);
    printf("{\"tw1\": %d, \"tw2\": %d, \"tw3\": %d, \"w512\": %d, \"logtab\": %d, \"mel_w\": %d, \"dct_w\": %d, \"total\": %d, \"mel_pad\": %d, \"dct_len\": %d, \"np_max\": %d,\n",
           L.tw1, L.tw2, L.tw3, L.w512, L.logtab, L.mel_w, L.dct_w, L.total, L.mel_pad, L.dct_len, L.np_max);
    printf("\"mel_start\": ["); for (int l = 0; l < 64; ++l) printf("%d%s", t.mel_start[l], l < 63 ? "," : "],\n");
    printf("\"pstart\": ["); for (int l = 0; l < 65; ++l) printf("%d%s", t.pstart[l], l < 64 ? "," : "],\n");
    printf("\"partner\": ["); for (int l = 0; l < 64; ++l) printf("%d%s", t.partner[l], l < 63 ? "," : "]}\n");
    return 0;
}
int main(int, char** argv) { return argv[1][1] == '6' ? run<double>(argv[2]) : run<float>(argv[2]); }
'''
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, 'dump.cpp'), 'w').write(src)
        mel_filterbank(16000, 20, 257).astype(np.float64).tofile(os.path.join(d, 'filters.bin'))
        subprocess.run(['g++', '-O1', '-std=c++17', '-I', os.path.join(REPO, 'mycroft_precise_amd', 'csrc'), os.path.join(d, 'dump.cpp'), '-o', os.path.join(d, 'dump')], check=True)
        out = subprocess.run([os.path.join(d, 'dump'), real, os.path.join(d, 'filters.bin')], capture_output=True, text=True, check=True).stdout
    return json.loads(out)


def kbase_of(l):
    return (l >> 4) + 4 * ((l >> 2) & 3) + 16 * (l & 3)


def frame_instructions(real, tab, xchg='stride5', power_skew=None):
    """-> list of (section, instruction kind, per-lane byte addresses)"""
    RS = 8 if real == 'f64' else 4
    CX = 2 * RS
    rd_cx = 'ds_read_b128' if RS == 8 else 'ds_read_b64'
    wr_cx = 'ds_write_b128' if RS == 8 else 'ds_write_b64'
    rd_r = 'ds_read_b64' if RS == 8 else 'ds_read_b32'
    wr_r = 'ds_write_b64' if RS == 8 else 'ds_write_b32'
    L = range(64)
    scratch_reals = 64 * 5 * 2
    img = 4 * scratch_reals * RS                 # table image behind the four waves' scratch (wave 0's scratch at 0)
    skip = tab['tw1']
    T = lambda off: img + off - skip
    if power_skew is None:
        power_skew = RS == 8
    ppos = (lambda k: k + (k >> 4)) if power_skew else (lambda k: k)

    def xi(lane, reg):                           # complex element of the exchange area -> byte address
        if xchg == 'stride5':
            return (lane * 5 + reg) * CX
        return (lane * 4 + (reg ^ ((lane >> 1) & 3))) * CX       # stride 4, slot XOR-swizzled by lane bits 2:1

    ins = []
    for k in range(3):
        ins.append(('pass a twiddles', rd_cx, [T(tab['tw1']) + (k * 64 + l) * CX for l in L]))
    for k in range(3):
        ins.append(('pass b twiddles', rd_cx, [T(tab['tw2']) + (k * 16 + (l & 15)) * CX for l in L]))
    for name, shift in (('exchange c', 2), ('exchange d', 0)):
        for r in range(4):
            ins.append((name + ' store', wr_cx, [xi(l, r) for l in L]))
        for rp in range(4):
            ins.append((name + ' load', rd_cx, [xi((l & ~(3 << shift)) | (rp << shift), (l >> shift) & 3) for l in L]))
        if shift == 2:
            for k in range(3):
                ins.append(('pass c twiddles', rd_cx, [T(tab['tw3']) + (k * 4 + (l & 3)) * CX for l in L]))
    for r in (0, 1):
        ins.append(('mirror store', wr_cx, [xi(l, r) for l in L]))
    for r in (1, 0):
        ins.append(('mirror load', rd_cx, [xi(tab['partner'][l], r) for l in L]))
    for j in range(2):
        ins.append(('W512 twiddles', rd_cx, [T(tab['w512']) + (j * 64 + l) * CX for l in L]))
    for j in range(4):
        bins = [[kbase_of(l), kbase_of(l) + 64, 256 - kbase_of(l), 192 - kbase_of(l)][j] for l in L]
        ins.append(('power store', wr_r, [ppos(b) * RS for b in bins]))
    ins.append(('power store (bin 128)', wr_r, [ppos(128) * RS if kbase_of(l) == 0 else None for l in L]))
    for i in range(tab['mel_pad']):
        ins.append(('mel: power load', rd_r, [(tab['mel_start'][l] + i) * RS for l in L]))
        ins.append(('mel: weight load', rd_r, [T(tab['mel_w']) + (i * 64 + l) * RS for l in L]))
    PART, LM = 288 * RS, 352 * RS
    ins.append(('partial sum store', wr_r, [PART + l * RS for l in L]))
    zero = 2 * (63 * 5 + 4) * RS
    for i in range(8):
        ins.append(('filter sums load', rd_r, [(PART + (tab['pstart'][l] + i) * RS if i < tab['pstart'][l + 1] - tab['pstart'][l] else zero) if l < 20 else None for l in L]))
    if RS == 8:
        ins.append(('log table load (data-dependent: modelled conflict-free)', 'ds_read_b128', [T(tab['logtab']) + l * 16 if (l < 20 or l == 63) else None for l in L]))
    ins.append(('log-mel store', wr_r, [LM + (20 if l == 63 else l) * RS if (l < 20 or l == 63) else None for l in L]))
    for i in range(tab['dct_len']):
        ins.append(('DCT: log-mel load', rd_r, [LM + (tab['dct_len'] * (l & 3) + i) * RS for l in L]))
        ins.append(('DCT: weight load', rd_r, [T(tab['dct_w']) + (i * 64 + l) * RS for l in L]))
    ins.append(('c0 load', rd_r, [LM + 20 * RS for l in L]))
    return ins


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--real', default='f64', choices=['f64', 'f32'])
    ap.add_argument('--xchg', default='stride5', choices=['stride5', 'swizzle'])
    ap.add_argument('--json', default='')
    args = ap.parse_args()
    tab = dump_tables(args.real)
    ins = frame_instructions(args.real, tab, args.xchg)
    by = {}
    for sec, kind, addrs in ins:
        cyc, ideal = array_cycles(kind, addrs)
        e = by.setdefault((sec, kind), [0, 0, 0, 0])
        e[0] += 1; e[1] += cyc; e[2] += ideal; e[3] += max(cyc, KINDS[kind][3])
    print('%-58s %-14s %5s %8s %8s %9s' % ('section', 'instruction', 'count', 'array', 'ideal', 'conflict'))
    tot = [0, 0, 0, 0]
    for (sec, kind), (n, cyc, ideal, eff) in by.items():
        print('%-58s %-14s %5d %8d %8d %9d' % (sec, kind, n, cyc, ideal, cyc - ideal))
        tot = [tot[0] + n, tot[1] + cyc, tot[2] + ideal, tot[3] + eff]
    print('%-58s %-14s %5d %8d %8d %9d   (%.1f %% of the array cycles are conflicts)' % ('TOTAL per frame (%s, exchange %s)' % (args.real, args.xchg), '', tot[0], tot[1], tot[2], tot[1] - tot[2],
                                                                                      100.0 * (tot[1] - tot[2]) / tot[1]))
    if args.json:
        json.dump({'real': args.real, 'xchg': args.xchg, 'instructions': tot[0], 'array_cycles': tot[1], 'ideal': tot[2]}, open(args.json, 'w'))


if __name__ == '__main__':
    main()
