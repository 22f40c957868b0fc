"""Test infrastructure: writes `.dbg` and `.column.annodbg` files in the layout csrc/boss_files.hpp reads — an independent
(Python) statement of the same container layouts (sdsl-lite's serialize() of int_vector / bit_vector / rrr_vector<63> /
sd_vector / wt_huff + byte_tree / rank_support_v, v5 / select_support_mcl; BOSS::serialize boss.cpp:286-336,
DBGSuccinct::serialize, ColumnCompressed::serialize annotate_column_compressed.cpp:179-205).  The rrr and sd encoders here are
checked against the files the reference wrote (tests/test_boss_files.py re-encodes their decoded content and compares bytes);
the rank / select supports are written with the sizes sdsl gives them and arbitrary content (the reader skips them)."""
import heapq
import struct
from math import comb

STATE_SMALL, STATE_DYN, STATE_STAT, STATE_FAST = 1, 2, 3, 4
CODE_RRR, CODE_SD, CODE_STAT = 0, 1, 2


def be(x):
    return struct.pack(">Q", x)


def le(x):
    return struct.pack("<Q", x)


def _words(big, bits):
    n = (bits + 63) // 64
    return b"".join(le((big >> (64 * i)) & (2 ** 64 - 1)) for i in range(n))


def int_vector(vals, width):
    big = 0
    for i, v in enumerate(vals):
        assert 0 <= v < (1 << width)
        big |= v << (i * width)
    bits = len(vals) * width
    return le(bits) + bytes([width]) + _words(big, bits)


def bit_vector(bits):
    big = 0
    for i, b in enumerate(bits):
        if b:
            big |= 1 << i
    return le(len(bits)) + _words(big, len(bits))


def int_vector64(n_words):
    return le(n_words * 64) + b"\0" * (8 * n_words)


def rank_support_v(n_bits):
    return int_vector64(((n_bits >> 9) + 1) * 2)


def rank_support_v5(n_bits):
    return int_vector64(((n_bits >> 11) + 1) * 2)


def select_support_mcl(args, n_bits):
    """args = the positions selected over (ascending).  arg_cnt; superblock (the position of every 4096th argument);
    mini_or_long; one int_vector per 4096 arguments: 64 relative positions of every 64th argument ("mini") or all 4096
    positions ("long").  Up to 64 arguments this is what sdsl writes, byte for byte (the reference-written fixtures);
    beyond, long and mini blocks simply alternate so that the reader skips both shapes."""
    cnt = len(args)
    out = le(cnt)
    if not cnt:
        return out
    sb = (cnt + 4095) >> 12
    logn = ((n_bits + 63) // 64 * 64).bit_length()          # hi(capacity) + 1
    out += int_vector([args[i << 12] for i in range(sb)], logn)
    kinds = [i % 2 == 0 for i in range(sb)] if sb > 1 else []      # True = mini blocks
    out += bit_vector(kinds)
    for i in range(sb):
        mine = args[i << 12:(i + 1) << 12]
        if not kinds or kinds[i]:
            rel = [mine[j] - mine[0] for j in range(0, len(mine), 64)]
            out += int_vector(rel + [0] * (64 - len(rel)), max(1, max(rel).bit_length()))
        else:
            out += int_vector(mine + [0] * (4096 - len(mine)), logn)
    return out


def _block_number(block_bits):
    """rrr_helper::bin_to_nr; a block with more ones than zeros: the number of its complement"""
    k = sum(block_bits)
    if 2 * k > 63:
        block_bits = [1 - b for b in block_bits]
        k = 63 - k
    nr = 0
    for p, b in enumerate(block_bits):
        if b:
            nr += comb(62 - p, k)
            k -= 1
    return nr


def rrr_vector63(bits):
    n = len(bits)
    n_blocks = n // 63 + 1
    bt, btnr, pos, ptrs, ranks, ones = [], 0, 0, [], [], 0
    for b in range(n_blocks):
        if b % 32 == 0:
            ptrs.append(pos)
            ranks.append(ones)
        blk = list(bits[b * 63:(b + 1) * 63])
        blk += [0] * (63 - len(blk))
        k = sum(blk)
        bt.append(k)
        ones += k
        c = comb(63, k)
        ln = 0 if c == 1 else c.bit_length()
        btnr |= _block_number(blk) << pos
        pos += ln
    ranks.append(ones)
    btnr_bits = [(btnr >> i) & 1 for i in range(max(pos, 64) if n_blocks == 1 else pos)]
    return (le(n) + int_vector(bt, 6) + bit_vector(btnr_bits)
            + int_vector(ptrs, max(1, pos.bit_length())) + int_vector(ranks, max(1, ones.bit_length())))


def sd_vector(size, ones):
    """sd_vector<>::serialize; `ones` ascending positions"""
    m = len(ones)
    if size == 0 and m == 0:               # a default-constructed sd_vector (the absent suffix-range index of a .dbg)
        return le(0) + b"\0" + le(0) + bytes([64]) + le(0) + le(0) + le(0)
    logm = max(m, 1).bit_length() if m else 1
    logn = max(size, 1).bit_length()
    if logm == logn:
        logm -= 1
    wl = max(1, logn - logm)
    high = [0] * (m + (1 << logm))
    low = []
    for i, p in enumerate(ones):
        high[(p >> wl) + i] = 1
        low.append(p & ((1 << wl) - 1))
    hb = bit_vector(high)
    return (le(size) + bytes([wl]) + int_vector(low, wl) + hb
            + select_support_mcl([i for i, b in enumerate(high) if b], len(high))
            + select_support_mcl([i for i, b in enumerate(high) if not b], len(high)))


def bit_vector_sd(bits):
    """bit_vector_sd::serialize (bit_vector_sd.hpp:273-277): the sparser of the vector and its complement + the flag"""
    ones = [i for i, b in enumerate(bits) if b]
    inverted = len(ones) > len(bits) // 2
    if inverted:
        ones = [i for i, b in enumerate(bits) if not b]
    return sd_vector(len(bits), ones) + bytes([1 if inverted else 0])


def bit_vector_stat(bits):
    ones = [i for i, b in enumerate(bits) if b]
    return bit_vector(bits) + be(len(ones)) + rank_support_v5(len(bits)) + select_support_mcl(ones, len(bits))


def adaptive(bits, code):
    body = {CODE_RRR: rrr_vector63, CODE_SD: bit_vector_sd, CODE_STAT: bit_vector_stat}[code](bits)
    return be(code) + body


def _huffman(counts):
    """a Huffman tree over the symbol counts: ("leaf", symbol) | ("inner", left, right)"""
    heap = [(c, s, ("leaf", s)) for s, c in sorted(counts.items())]
    heapq.heapify(heap)
    tick = 1000
    while len(heap) > 1:
        a, b = heapq.heappop(heap), heapq.heappop(heap)
        heapq.heappush(heap, (a[0] + b[0], tick, ("inner", a[2], b[2])))
        tick += 1
    return heap[0][2]


def _members(t):
    return {t[1]} if t[0] == "leaf" else _members(t[1]) | _members(t[2])


def wt_huff(seq, rrr):
    """wt_pc::serialize with a byte_tree: nodes numbered breadth first from the root, every inner node's stretch of the level
    bit vector in that order, a leaf keeping its symbol in bv_pos_rank"""
    counts = {}
    for s in seq:
        counts[s] = counts.get(s, 0) + 1
    # breadth-first: (tree, parent index, the subsequence routed to it)
    nodes, queue = [], [(_huffman(counts), 0xFFFF, list(seq))]
    while queue:
        t, parent, sub = queue.pop(0)
        idx = len(nodes)
        nodes.append({"t": t, "parent": parent, "sub": sub, "child": [0xFFFF, 0xFFFF]})
        if parent != 0xFFFF:
            ch = nodes[parent]["child"]
            ch[0 if ch[0] == 0xFFFF else 1] = idx
        if t[0] == "inner":
            right = _members(t[2])
            queue.append((t[1], idx, [s for s in sub if s not in right]))
            queue.append((t[2], idx, [s for s in sub if s in right]))
    bits = []
    for nd in nodes:
        nd["bv_pos"] = len(bits)
        if nd["t"][0] == "inner":
            right = _members(nd["t"][2])
            bits += [1 if s in right else 0 for s in nd["sub"]]
    ones_before = [0]
    for b in bits:
        ones_before.append(ones_before[-1] + b)
    out = le(len(seq)) + le(len(counts))
    if rrr:
        out += rrr_vector63(bits)
    else:
        out += (bit_vector(bits) + rank_support_v(len(bits)) + select_support_mcl([i for i, b in enumerate(bits) if b], len(bits))
                + select_support_mcl([i for i, b in enumerate(bits) if not b], len(bits)))
    out += le(len(nodes))
    c_to_leaf, path = [0xFFFF] * 256, [0] * 256
    for i, nd in enumerate(nodes):
        leaf = nd["t"][0] == "leaf"
        out += le(nd["bv_pos"]) + le(nd["t"][1] if leaf else ones_before[nd["bv_pos"]]) + struct.pack("<HHH", nd["parent"], *nd["child"])
        if leaf:
            c_to_leaf[nd["t"][1]] = i
            chain, v = [], i
            while nodes[v]["parent"] != 0xFFFF:
                chain.append(1 if nodes[nodes[v]["parent"]]["child"][1] == v else 0)
                v = nodes[v]["parent"]
            word = 0
            for ln, b in enumerate(reversed(chain)):
                word |= b << ln
            path[nd["t"][1]] = word | (len(chain) << 56)
    return out + struct.pack("<256H", *c_to_leaf) + struct.pack("<256Q", *path)


def dbg_file(k, W, last, F, mode=0, state=STATE_SMALL, last_code=CODE_RRR, suffix_index=True):
    """BOSS::serialize + DBGSuccinct::serialize; k = the DBG's k"""
    W, last = [int(x) for x in W], [int(x) for x in last]
    logsigma = (len(F) - 1).bit_length() + 1
    out = be(len(F)) + b"".join(be(int(x)) for x in F) + be(k - 1) + be(state)
    if state == STATE_SMALL:
        out += wt_huff(W, True) + be(logsigma) + adaptive(last, last_code)
    elif state == STATE_STAT:
        out += wt_huff(W, False) + be(logsigma) + bit_vector_stat(last)
    elif state == STATE_FAST:
        out += int_vector(W, logsigma)
        for c in range(1 << logsigma):
            out += bit_vector_stat([1 if w == c else 0 for w in W])
        out += bit_vector_stat(last)
    else:
        out += b"\0" * 64
    out += be(mode)
    if suffix_index:                       # BOSS::serialize_suffix_ranges with no index: length 0 + an empty sd_vector
        out += be(0) + sd_vector(0, [])
    return out


def label_encoder_v1(labels):
    enc = lambda s: bytes([len(s)]) + s.encode() if len(s) < 0x80 else bytes([0xC0 | (len(s) >> 6), 0x80 | (len(s) & 0x3F)]) + s.encode()
    out = be(len(labels)) + b"".join(enc(s) for s in labels)
    out += int_vector(list(range(len(labels))), 64)
    out += be(len(labels)) + b"".join(enc(s) for s in labels)
    return out


def label_encoder_v2(labels):
    n_buckets = 1
    while n_buckets < 2 * max(1, len(labels)):
        n_buckets *= 2
    out = b"LE-v2.0" + le(1) + le(len(labels)) + le(n_buckets) + struct.pack("<f", 0.75)
    out += b"".join(be(len(s)) + s.encode() for s in labels)
    out += b"".join(le(i) + le(0x1234 + i) for i in range(n_buckets))
    return out


def column_file(n_rows, labels, columns, codes=None, v2=False):
    """columns[j] = set of rows with label j"""
    out = be(n_rows) + (label_encoder_v2 if v2 else label_encoder_v1)(labels)
    for j, col in enumerate(columns):
        s = set(int(r) for r in col)
        bits = [1 if r in s else 0 for r in range(n_rows)]
        out += adaptive(bits, codes[j] if codes else CODE_SD)
    return out


def edgemask_file(valid, state):
    """the `.edgemask` of DBGSuccinct::serialize: a bit_vector_stat next to a FAST-state graph, a bit_vector_small otherwise
    (the smaller of sd and rrr; here: sd when few edges are masked or few are valid)"""
    bits = [int(b) for b in valid]
    if state == STATE_FAST:
        return bit_vector_stat(bits)
    ones = sum(bits)
    return adaptive(bits, CODE_SD if min(ones, len(bits) - ones) * 8 < len(bits) else CODE_RRR)
