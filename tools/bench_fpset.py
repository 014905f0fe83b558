#!/usr/bin/env python3
"""Micro-benchmark of the stand-alone seen-set (vsrmc_fpset_*, the drop-in for tlc2.tool.fp.FPSet.putBlock / containsBlock):
SURVEY §8(d) inputs — 2^28 uniform 64-bit keys from splitmix64(seed 0x5EED) with 85 % duplicates, into tables that end at
load factor 0.5 and 0.8.  Keys are generated on the GPU with torch and handed over as device pointers (batches of 2^24, the
JNI shim's unit).  Prints one JSON line per case: keys/s, algorithmic GB/s (8 B key read + 1 B verdict written per key,
16 B slot written per insert), HBM-sector GB/s (64 B per probe)."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vsr_tlaplus_amd as vt  # noqa: E402


def splitmix64(x):
    """x: int64 tensor (bit pattern of a u64) -> int64 tensor"""
    def mul(a, c):
        return a * torch.tensor(c - (1 << 64) if c >= (1 << 63) else c, dtype=torch.int64, device=a.device)

    def shr(a, k):      # logical shift right on the u64 bit pattern
        return (a >> k) & ((1 << (64 - k)) - 1)
    x = x + torch.tensor(0x9E3779B97F4A7C15 - (1 << 64), dtype=torch.int64, device=x.device)
    z = x
    z = mul(z ^ shr(z, 30), 0xBF58476D1CE4E5B9)
    z = mul(z ^ shr(z, 27), 0x94D049BB133111EB)
    return z ^ shr(z, 31)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log2-keys", type=int, default=28)
    ap.add_argument("--dup", type=float, default=0.85)
    ap.add_argument("--batch-log2", type=int, default=24)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    n = 1 << a.log2_keys
    unique = int(n * (1.0 - a.dup))
    g = torch.Generator(device=dev)
    g.manual_seed(0x5EED)
    ids = torch.randint(0, unique, (n,), generator=g, device=dev, dtype=torch.int64)
    ids[:unique] = torch.arange(unique, device=dev, dtype=torch.int64)[torch.randperm(unique, generator=g, device=dev)]
    keys = splitmix64(ids + 0x5EED)
    keys = torch.where(keys == 0, torch.ones_like(keys), keys)
    del ids
    out = torch.empty(n, dtype=torch.uint8, device=dev)
    for load in (0.5, 0.8):
        import math
        log2_slots = max(8, int(math.ceil(math.log2(unique / load))))
        eff_load = unique / float(1 << log2_slots)
        s = vt.FPSet(log2_slots=log2_slots, device=0)
        b = 1 << a.batch_log2
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(0, n, b):
            k = min(b, n - i)
            s.put_block_device(keys[i:i + k].data_ptr(), k, out[i:i + k].data_ptr())
        torch.cuda.synchronize()
        t_put = time.perf_counter() - t0
        new = int((out == 0).sum().item())
        assert new == unique == s.size(), (new, unique, s.size())
        t0 = time.perf_counter()
        for i in range(0, n, b):
            k = min(b, n - i)
            s.contains_block_device(keys[i:i + k].data_ptr(), k, out[i:i + k].data_ptr())
        torch.cuda.synchronize()
        t_get = time.perf_counter() - t0
        assert int(out.sum().item()) == n
        alg_put = (9.0 * n + 16.0 * unique) / t_put / 1e9
        print(json.dumps(dict(case="put", keys=n, unique=unique, slots_log2=log2_slots, final_load=round(eff_load, 3),
                              keys_per_s=round(n / t_put, 1), seconds=round(t_put, 4), alg_GBps=round(alg_put, 1),
                              sector_GBps=round(64.0 * n / t_put / 1e9, 1))))
        print(json.dumps(dict(case="contains", keys=n, slots_log2=log2_slots, final_load=round(eff_load, 3),
                              keys_per_s=round(n / t_get, 1), seconds=round(t_get, 4), alg_GBps=round(9.0 * n / t_get / 1e9, 1),
                              sector_GBps=round(64.0 * n / t_get / 1e9, 1))))
        s.close()


if __name__ == "__main__":
    main()
