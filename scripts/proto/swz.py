import itertools
G128 = [[0,1,2,3,12,13,14,15,20,21,22,23,24,25,26,27],[4,5,6,7,8,9,10,11,16,17,18,19,28,29,30,31]]
G128 = G128 + [[l+32 for l in g] for g in G128]
def read_conf(addr_of_lane):
    worst = 1
    for g in G128:
        banks = {}
        for l in g:
            a = addr_of_lane(l)
            slot = (a // 16) % 16
            banks.setdefault(slot, set()).add(a)
        worst = max(worst, max(len(v) for v in banks.values()))
    return worst
def write_conf(addr_of_lane):
    worst = 1
    for g0 in range(0, 64, 8):
        banks = {}
        for l in range(g0, g0 + 8):
            a = addr_of_lane(l)
            slot = (a // 16) % 8
            banks.setdefault(slot, set()).add(a)
        worst = max(worst, max(len(v) for v in banks.values()))
    return worst
cands = {
 'p>>1': lambda p: (p >> 1) & 7,
 'p': lambda p: p & 7,
 'p+(p>>3)': lambda p: (p + (p >> 3)) & 7,
 'p^(p>>3)': lambda p: (p ^ (p >> 3)) & 7,
 '(p>>1)^((p&1)*4)': lambda p: ((p >> 1) ^ ((p & 1) * 4)) & 7,
 '(p>>1)+(p&1)*4': lambda p: ((p >> 1) + (p & 1) * 4) & 7,
 '(p>>1)^(p<<2)': lambda p: ((p >> 1) ^ (p << 2)) & 7,
}
for name, g in cands.items():
    rw = 1
    for dx in range(3):
        for ch0 in range(8):
            f = lambda l: (((l & 31) + dx) * 128) + (((ch0 ^ (l >> 5)) ^ g((l & 31) + dx)) << 4)
            rw = max(rw, read_conf(f))
    ww = 1
    for ch0 in range(8):
        f = lambda l: (((l & 31) + 1) * 128) + (((ch0 ^ 2 * (l >> 5)) ^ g((l & 31) + 1)) << 4)
        ww = max(ww, write_conf(f))
    print(name, 'read', rw, 'write', ww)
# padded record, no swizzle
for rec in (144, 160, 272):
    rw = 1
    for dx in range(3):
        for ch0 in range(4):
            f = lambda l: ((l & 31) + dx) * rec + (2 * ch0 + (l >> 5)) * 16
            rw = max(rw, read_conf(f))
    print('rec', rec, 'read', rw)
