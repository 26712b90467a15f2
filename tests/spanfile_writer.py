"""Test-side WRITER of the reference's spanfile format (spanfile.go:1-22 grammar,
:568-661 7-code, :679-728 serializeSpan, :836-849 CRC32-IEEE trailer), used to make
collection files for the pager tests -- there is no Go toolchain to make real ones.
Pinned against the byte-level worked examples of SURVEY.md Appendix A
(tests/test_pager_cpu.py::test_writer_matches_survey_examples)."""
import json
import struct
import zlib

ACTIVE = 0x5350414E  # 'SPAN'
FREE = 0x46524545    # 'FREE'


def write7(n):
    """write7Code, spanfile.go:568-625 (note the `<` thresholds: 127 takes two bytes)."""
    limits = [0x7F, 0x3FFF, 0x1FFFFF, 0xFFFFFFF, 0x7FFFFFFFF, 0x3FFFFFFFFFF, 0x1FFFFFFFFFFFF,
              0xFFFFFFFFFFFFFF]
    width = 9
    for i, lim in enumerate(limits):
        if n < lim:
            width = i + 1
            break
    out = bytearray()
    for k in range(width - 1, 0, -1):
        out.append(((n >> (7 * k)) & 0x7F) | 0x80)
    out.append(n & 0x7F)
    return bytes(out)


def span(seq, record_id, streams, magic=ACTIVE, pad=0, corrupt=False):
    """One span: streams = [(stream_id, bytes), ...]; pad zero bytes before the CRC."""
    rid = record_id.encode() if isinstance(record_id, str) else record_id
    body = write7(seq) + write7(len(rid)) + rid + bytes([len(streams)])
    for sid, data in streams:
        body += bytes([sid]) + write7(len(data)) + data
    length = 8 + len(body) + pad + 4
    buf = struct.pack(">II", magic, length) + body + b"\x00" * pad
    crc = zlib.crc32(buf) & 0xFFFFFFFF
    if corrupt:
        crc ^= 0x5A5A5A5A
    return buf + struct.pack(">I", crc)


def header_span(seq, name, metric, dim, bits):
    js = json.dumps({"name": name, "distance_method": metric, "dimension_count": dim,
                     "quantization": bits}, separators=(",", ":")).encode()
    return span(seq, "", [(0, js)])


def collection_file(path, metric, dim, bits, docs, extra_spans=(), tail_zeros=4096):
    """docs: iterable of (id, metadata bytes, packed vector bytes)."""
    seq = 1
    blob = span(0, "", [], magic=FREE)  # the initial empty span, freed once the header is written
    blob += header_span(seq, str(path), metric, dim, bits)
    for id, meta, vec in docs:
        seq += 1
        blob += span(seq, str(id), [(0, meta), (1, vec)])
    for s in extra_spans:
        blob += s
    blob += b"\x00" * tail_zeros  # files grow by zero bytes; magic 0 = rest is free
    with open(path, "wb") as f:
        f.write(blob)
    return seq
