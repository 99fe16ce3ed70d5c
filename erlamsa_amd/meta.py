"""Meta trace of a case (EH_FLAG_META_TRACE, include/erlamsa_hip.h eh_result_meta): decoder of the engine's event bytes and
renderer to the text erlamsa's meta logger writes (reference src/erlamsa_main.erl:58-70: every element of
lists:reverse(lists:flatten(Meta)) with io_lib:format("~p~n", [X])).

decode(buf, atoms) -> [term]; a term is ("atom", name) | int | float | ("str", bytes) | tuple of terms.
render(term) -> the ~p text of one element; lines(terms) -> the block the logger prints for a case.
host_terms(...) adds the entries that the host side of the reference conses around the batch path ({nth, I}, the generator's,
{output, return}, {written, N}: erlamsa_main.erl:185-195, erlamsa_gen.erl:102-164, erlamsa_out.erl:97).
"""

K_AA, K_AI, K_SIZER, K_CSUM, K_SKIPPED, K_ARCHIVER = 1, 2, 3, 4, 5, 6

PATTERN_CODE = {"once_dec": "od", "many_dec": "nd", "burst": "bu", "skipper": "sk", "sizer": "sz", "csum": "cs", "archiver": "ar",
                "compressed": "cp", "no_muta": "nu"}


class Truncated(Exception):
    pass


def _varint(buf, p):
    v = s = 0
    while True:
        if p >= len(buf):
            raise Truncated()
        b = buf[p]; p += 1
        v |= (b & 127) << s
        s += 7
        if not b & 128:
            return v, p


def decode(buf, atoms):
    """-> (terms, truncated)"""
    buf = bytes(buf)
    cut = len(buf) == 32768 and buf[-1] == 0xFF
    if cut:
        buf = buf[:-1]
    out, p = [], 0
    A = lambda i: ("atom", atoms[i])
    try:
        while p < len(buf):
            k = buf[p]; q = p + 1
            if k == K_AA:
                if q + 2 > len(buf):
                    raise Truncated()
                t = (A(buf[q]), A(buf[q + 1])); q += 2
            elif k == K_AI:
                if q + 1 > len(buf):
                    raise Truncated()
                a = A(buf[q]); z, q = _varint(buf, q + 1)
                t = (a, (z >> 1) ^ -(z & 1))
            elif k == K_SIZER:
                if q + 2 > len(buf):
                    raise Truncated()
                size, big = buf[q] * 8, buf[q + 1]; ln, q = _varint(buf, q + 2); a, q = _varint(buf, q); b, q = _varint(buf, q)
                t = (("atom", "sizer"), (("atom", "ok"), size, ("atom", "big" if big else "little"), ln, a, b))
            elif k == K_CSUM:
                if q + 1 > len(buf):
                    raise Truncated()
                crc = buf[q]; pl, q = _varint(buf, q + 1); bl, q = _varint(buf, q)
                t = (("atom", "csum"), (("atom", "crc32" if crc else "xor8"), 32 if crc else 8, pl, bl))
            elif k == K_SKIPPED:
                n, q = _varint(buf, q)
                t = (("atom", "skipped"), float(n))
            elif k == K_ARCHIVER:
                n, q = _varint(buf, q)
                if q + n > len(buf):
                    raise Truncated()
                t = (("atom", "archiver"), ("str", buf[q:q + n])); q += n
            else:
                raise ValueError("unknown meta event kind %d at byte %d" % (k, p))
            out.append(t); p = q
    except Truncated:
        if not cut:
            raise ValueError("meta trace ends inside an event")
    return out, cut


def _float_p(f):
    """io_lib_format:fwrite_g/1 for the floats the path makes (whole numbers >= 0): shortest digits, plain while that is not longer
    than the exponent form"""
    k = int(f)
    if k == 0:
        return "0.0"
    digs = str(k); place = len(digs); digs = digs.rstrip("0") or "0"
    l, exp = len(digs), place - 1
    if place - l <= len(str(exp)) + 2:
        return digs + "0" * (place - l) + ".0"
    return digs[0] + "." + (digs[1:] or "0") + "e" + str(exp)


def _printable(ch):
    return 32 <= ch <= 126 or ch in (8, 9, 10, 11, 12, 13, 27) or ch >= 160


def render(t):
    if isinstance(t, bool):
        return "true" if t else "false"
    if isinstance(t, int):
        return str(t)
    if isinstance(t, float):
        return _float_p(t)
    if t[0] == "atom" and len(t) == 2 and isinstance(t[1], str):
        return t[1]
    if t[0] == "str" and len(t) == 2 and isinstance(t[1], (bytes, bytearray)):
        s = bytes(t[1])
        if not s:
            return "[]"
        if not all(_printable(ch) for ch in s):
            return "[" + ",".join(str(ch) for ch in s) + "]"
        esc = {34: '\\"', 92: "\\\\", 10: "\\n", 13: "\\r", 9: "\\t", 11: "\\v", 8: "\\b", 12: "\\f", 27: "\\e"}
        return '"' + "".join(esc.get(ch, chr(ch)) for ch in s) + '"'
    return "{" + ",".join(render(x) for x in t) + "}"


def lines(terms):
    return "".join(render(t) + "\n" for t in terms)


def legacy(terms, mutator_names):
    """the short form of earlier ABI versions: (kind, name) for {failed | used, Mutator}, {pattern, _} (pattern codes), {skipped_big, _}"""
    out = []
    for t in terms:
        a = t[0][1] if isinstance(t[0], tuple) and t[0][0] == "atom" else None
        if a in ("failed", "used") and isinstance(t[1], tuple) and t[1][0] == "atom" and t[1][1] in mutator_names:
            out.append((a, t[1][1]))
        elif a == "pattern" and t[1][1] in PATTERN_CODE:
            out.append(("pattern", PATTERN_CODE[t[1][1]]))
        elif a == "skipped_big":
            out.append(("skipped_big", ""))
    return out


def host_terms(nth, generator, terms, written, output="return"):
    """The list the reference's RecordMeta gets for a case: [{nth, I}, GenMeta] and the output's entry in front of what Pat built,
    {written, N} behind it (erlamsa_main.erl:185-195; GenMeta: {generator, direct} | {generator, random} | [{generator, file | jump},
    {source, path}] erlamsa_gen.erl:115,145,164,178)."""
    gm = [(("atom", "generator"), ("atom", generator))] + ([(("atom", "source"), ("atom", "path"))] if generator in ("file", "jump") else [])
    # GenMeta is consed as ONE element [{generator, G}, {source, path}]: flatten keeps its order, the final reverse turns it round
    return [(("atom", "nth"), nth)] + gm[::-1] + [(("atom", "output"), ("atom", output))] + list(terms) + [(("atom", "written"), written)]
