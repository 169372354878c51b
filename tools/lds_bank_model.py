"""Developer tool: LDS bank-conflict model (MI355X guide, ds_read_b128 / ds_write_b128 lane groups) for the row-sum ring of
k_level_pass's blur: cycles per wave instruction for candidate lane mappings, slot layouts and slot strides."""
R_GROUPS = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)), list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
            list(range(32, 36)) + list(range(44, 48)) + list(range(52, 60)), list(range(36, 44)) + list(range(48, 52)) + list(range(60, 64))]
W_GROUPS = [list(range(8 * i, 8 * i + 8)) for i in range(8)]


def cycles(addrs, groups):
    tot = 0
    for g in groups:
        banks = {}
        for l in g:
            a = addrs[l]
            if a is None:
                continue
            for d in range(4):
                banks.setdefault(((a // 4) + d) % 64, set()).add(a + 4 * d)
        tot += max([len(v) for v in banks.values()] + [1])
    return tot


if __name__ == "__main__":
    for nsegs in (3, 4, 5, 6):
        R = min(28, 64 // nsegs)
        res = []
        for mapping in ("seg_minor", "rp_minor"):
            for layout in ("interleaved", "half_major"):
                for pad in range(0, 5):
                    stride = nsegs * 32 + 16 * pad
                    wr = ww = 0
                    for s0 in range(32):
                        for half in (0, 1):
                            rd = []
                            for lane in range(64):
                                rp, seg = divmod(lane, nsegs) if mapping == "seg_minor" else divmod(lane, R)[::-1]
                                if rp >= R or seg >= nsegs:
                                    rd.append(None)
                                    continue
                                off = seg * 32 + half * 16 if layout == "interleaved" else half * nsegs * 16 + seg * 16
                                rd.append(((s0 + rp) % 32) * stride + off)
                            wr, ww = max(wr, cycles(rd, R_GROUPS)), max(ww, cycles(rd, W_GROUPS))
                    res.append((wr, ww, stride * 32, mapping, layout, stride))
        res.sort()
        print(f"{nsegs} segments, {R} pairs per step: (read cycles [ideal 4], write cycles [ideal 8], ring bytes, mapping, layout, stride)", res[:4])
