"""sha1 of every gfx950 code object embedded in a built library (default: moshi_amd/libmoshi_mi.so).

    python scripts/device_code_hash.py [lib.so ...]

Used to show that a host-side change (error codes, test hooks) leaves the GPU code that a committed GPU run validated untouched:
profiles/r05_logs/device_code_sha1.txt holds the hashes of the library the round's last GPU suite ran on."""
import hashlib
import re
import struct
import sys
from pathlib import Path


def code_objects(path):
    data = Path(path).read_bytes()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    out = []
    for m in re.finditer(magic, data):
        i = m.start()
        p = i + len(magic)
        num = struct.unpack_from("<Q", data, p)[0]
        p += 8
        for _ in range(num):
            off, size, tlen = struct.unpack_from("<QQQ", data, p)
            p += 24
            triple = data[p:p + tlen].decode()
            p += tlen
            if "gfx950" in triple and size > 0:
                out.append((triple, size, hashlib.sha1(data[i + off:i + off + size]).hexdigest()))
    return out


if __name__ == "__main__":
    libs = sys.argv[1:] or [str(Path(__file__).resolve().parent.parent / "moshi_amd" / "libmoshi_mi.so")]
    for lib in libs:
        for triple, size, h in code_objects(lib):
            print(f"{h}  {size:9d}  {triple}")
