// Bristol / Bristol Fashion text -> rv_op stream (host only).  See include/reverie_amd.h.
// The reference delegates this to the un-vendored `mcircuit` crate (README.md:14-16,
// src/lib.rs:6-7); the format is the public one.
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "../../include/reverie_amd.h"

namespace {

struct Tok {
    const char* p;
    const char* end;
    bool next_line(std::vector<std::string>& out) {  // skips blank lines
        out.clear();
        while (p < end) {
            const char* e = (const char*)memchr(p, '\n', (size_t)(end - p));
            if (!e) e = end;
            const char* q = p;
            while (q < e) {
                while (q < e && (*q == ' ' || *q == '\t' || *q == '\r')) q++;
                const char* s = q;
                while (q < e && *q != ' ' && *q != '\t' && *q != '\r') q++;
                if (q > s) out.emplace_back(s, q);
            }
            p = e < end ? e + 1 : end;
            if (!out.empty()) return true;
        }
        return false;
    }
};

// A decimal number.  The reference's wire ids are `usize` (interpreter/single.rs:14,109-155); rv_op carries u32, so a file that
// names a wire or a count of 2^32 and more is well-formed but beyond this library: `too_big` is set and the parser answers
// RV_E_UNSUPPORTED instead of calling the file malformed.
bool g_too_big_dummy;
bool to_u64(const std::string& s, uint64_t& v, bool& too_big = g_too_big_dummy) {
    if (s.empty()) return false;
    v = 0;
    bool big = false;
    for (char c : s) {
        if (c < '0' || c > '9') return false;
        if (!big) v = v * 10 + (uint64_t)(c - '0');
        if (v > 0xFFFFFFFFull) big = true;
    }
    if (big) too_big = true;
    return !big;
}

rv_op mk(uint8_t opcode, uint32_t dst, uint32_t a, uint32_t b, uint64_t imm) {
    rv_op o;
    memset(&o, 0, sizeof o);
    o.domain = RV_DOM_GF2;
    o.opcode = opcode;
    o.dst = dst;
    o.a = a;
    o.b = b;
    o.imm = imm;
    return o;
}

}  // namespace

static int parse_impl(const char* text, size_t len, int format, const uint8_t* expected_outputs, size_t n_expected, rv_op** ops,
                      size_t* n_ops, rv_bristol_info* info);

extern "C" int rv_bristol_parse(const char* text, size_t len, int format, const uint8_t* expected_outputs, size_t n_expected,
                                rv_op** ops, size_t* n_ops, rv_bristol_info* info) {
    try {  // no C++ exception may cross the C boundary
        return parse_impl(text, len, format, expected_outputs, n_expected, ops, n_ops, info);
    } catch (...) {
        return RV_E_NOMEM;
    }
}

static int parse_impl(const char* text, size_t len, int format, const uint8_t* expected_outputs, size_t n_expected, rv_op** ops,
                      size_t* n_ops, rv_bristol_info* info) {
    if (!text || !ops || !n_ops) return RV_E_ARG;
    *ops = nullptr;
    *n_ops = 0;
    Tok t{text, text + len};
    bool too_big = false;  // a number of 2^32 or more was read: RV_E_UNSUPPORTED rather than "malformed"
#define BAD() return (too_big ? RV_E_UNSUPPORTED : RV_E_BAD_OP)
    std::vector<std::string> l1, l2, l3;
    uint64_t n_gates, n_wires;
    if (!t.next_line(l1) || l1.size() != 2 || !to_u64(l1[0], n_gates, too_big) || !to_u64(l1[1], n_wires, too_big)) BAD();
    if (!t.next_line(l2)) BAD();
    uint64_t n_in = 0, n_out = 0;
    uint64_t first = 0;
    if (!to_u64(l2[0], first, too_big)) BAD();
    bool fashion = format == 1 || (format == 0 && l2.size() == first + 1 && !(l2.size() == 3 && first != 2));
    if (format == 0 && l2.size() == 3 && first == 2) {
        // ambiguous "2 a b": Fashion has a third header line "nov n..", old Bristol goes straight to gates
        Tok probe = t;
        std::vector<std::string> peek;
        fashion = probe.next_line(peek) && !peek.empty() && peek.size() >= 2 && peek.back().find_first_not_of("0123456789") == std::string::npos;
    }
    if (fashion) {
        if (l2.size() != first + 1) BAD();
        for (size_t i = 1; i < l2.size(); i++) {
            uint64_t v;
            if (!to_u64(l2[i], v, too_big)) BAD();
            n_in += v;
        }
        uint64_t nov;
        if (!t.next_line(l3) || !to_u64(l3[0], nov, too_big) || l3.size() != nov + 1) BAD();
        for (size_t i = 1; i < l3.size(); i++) {
            uint64_t v;
            if (!to_u64(l3[i], v, too_big)) BAD();
            n_out += v;
        }
    } else {
        if (l2.size() != 3) BAD();
        uint64_t a, b, c;
        if (!to_u64(l2[0], a, too_big) || !to_u64(l2[1], b, too_big) || !to_u64(l2[2], c, too_big)) BAD();
        n_in = a + b;
        n_out = c;
    }
    if (n_in > n_wires || n_out > n_wires) return RV_E_WIRE_OOB;
    // the assertions read expected_outputs[0 .. n_out): the caller states how many it passed, checked before any is read
    if (expected_outputs && n_expected != n_out) return RV_E_ARG;

    rv_bristol_info bi;
    memset(&bi, 0, sizeof bi);
    bi.n_gates = n_gates;
    bi.n_wires = n_wires;
    bi.n_inputs = n_in;
    bi.n_outputs = n_out;
    std::vector<rv_op> out;
    if (n_gates > len / 8 + 1) BAD();  // every gate line needs at least 8 characters
    // (a header may claim billions of input wires in a few bytes of text: reserve what the TEXT can justify, let the
    // vector grow for the rest -- a failed allocation is caught by the caller and reported as RV_E_NOMEM)
    out.reserve((size_t)std::min<uint64_t>(n_in + n_gates + 2 * n_out, (uint64_t)len + 4096));
    for (uint64_t w = 0; w < n_in; w++) out.push_back(mk(RV_OP_INPUT, (uint32_t)w, 0, 0, 0));
    std::vector<std::string> g;
    std::vector<uint64_t> v;
    for (uint64_t gi = 0; gi < n_gates; gi++) {
        if (!t.next_line(g) || g.size() < 4) BAD();
        const std::string& kind = g.back();
        v.assign(g.size() - 1, 0);
        for (size_t i = 0; i + 1 < g.size(); i++)
            if (!to_u64(g[i], v[i], too_big)) BAD();
        const uint64_t nin = v[0], nout = v[1];
        if (v.size() != 2 + nin + nout) BAD();
        const uint64_t* in = v.data() + 2;
        const uint64_t* o = in + nin;
        auto wire_ok = [&](uint64_t w) { return w < n_wires; };
        if (kind == "XOR" || kind == "AND") {
            if (nin != 2 || nout != 1 || !wire_ok(in[0]) || !wire_ok(in[1]) || !wire_ok(o[0])) return kind.empty() ? RV_E_BAD_OP : (nin != 2 || nout != 1 ? RV_E_BAD_OP : RV_E_WIRE_OOB);
            out.push_back(mk(kind == "XOR" ? RV_OP_ADD : RV_OP_MUL, (uint32_t)o[0], (uint32_t)in[0], (uint32_t)in[1], 0));
            (kind == "XOR" ? bi.n_xor : bi.n_and)++;
        } else if (kind == "INV" || kind == "NOT") {
            if (nin != 1 || nout != 1) BAD();
            if (!wire_ok(in[0]) || !wire_ok(o[0])) return RV_E_WIRE_OOB;
            out.push_back(mk(RV_OP_ADDCONST, (uint32_t)o[0], (uint32_t)in[0], 0, 1));
            bi.n_inv++;
        } else if (kind == "EQW") {
            if (nin != 1 || nout != 1) BAD();
            if (!wire_ok(in[0]) || !wire_ok(o[0])) return RV_E_WIRE_OOB;
            out.push_back(mk(RV_OP_ADDCONST, (uint32_t)o[0], (uint32_t)in[0], 0, 0));
            bi.n_other++;
        } else if (kind == "EQ") {  // constant assignment: the "input" is the literal 0/1
            if (nin != 1 || nout != 1 || in[0] > 1) BAD();
            if (!wire_ok(o[0])) return RV_E_WIRE_OOB;
            out.push_back(mk(RV_OP_CONST, (uint32_t)o[0], 0, 0, in[0]));
            bi.n_other++;
        } else if (kind == "MAND") {
            if (nin != 2 * nout || nout == 0) BAD();
            for (uint64_t k = 0; k < nout; k++) {
                if (!wire_ok(in[k]) || !wire_ok(in[nout + k]) || !wire_ok(o[k])) return RV_E_WIRE_OOB;
                out.push_back(mk(RV_OP_MUL, (uint32_t)o[k], (uint32_t)in[k], (uint32_t)in[nout + k], 0));
                bi.n_and++;
            }
        } else {
            BAD();
        }
    }
    uint64_t wires = n_wires;
    if (expected_outputs) {
        for (uint64_t k = 0; k < n_out; k++) {
            const uint32_t w = (uint32_t)(n_wires - n_out + k);
            const uint32_t tmp = (uint32_t)wires++;
            out.push_back(mk(RV_OP_ADDCONST, tmp, w, 0, expected_outputs[k] & 1));
            out.push_back(mk(RV_OP_ASSERTZERO, 0, tmp, 0, 0));
        }
    }
    bi.gf2_wires = wires;
    if (wires > 0xFFFFFFFFull) return RV_E_UNSUPPORTED;
    rv_op* res = (rv_op*)malloc(sizeof(rv_op) * (out.empty() ? 1 : out.size()));
    if (!res) return RV_E_NOMEM;
    if (!out.empty()) memcpy(res, out.data(), sizeof(rv_op) * out.size());
    *ops = res;
    *n_ops = out.size();
    if (info) *info = bi;
    return RV_OK;
}
