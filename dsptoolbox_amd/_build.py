"""Build the HIP library in-tree:  python -m dsptoolbox_amd._build [--force]."""

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libdsptoolbox_amd.so")
SOURCES = [os.path.join(CSRC, "api.hip")]
# -fno-slp-vectorize: hipcc's SLP pass turns the complex butterflies into v_pk_* pairs glued
# together with ~270 v_mov per FFT; scalar fp32 VALU code is 25 % faster here (measured on
# MI355X, profiles/).
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-shared", "-fPIC"]
# per-kernel registers / scratch / LDS as the compiler reports them (parsed into RES_PATH)
REMARK_FLAGS = ["-Rpass-analysis=kernel-resource-usage"]
RES_PATH = os.path.join(LIB_DIR, "kernel_resources.json")
# Kernels whose pair / block loops must not touch scratch memory: the build FAILS if one of them
# is compiled with ScratchSize > 0 (a spill inside a register-resident transform costs more than
# the occupancy it buys).  Matched as substrings of the demangled-ish mangled names.
NO_SCRATCH = [
    "welch40964k_y3", "welch40964k_x3", "welch40963k_y", "welch40963k_x",
    "stft1k11k_stft_wave", "stft4k6k_stft", "stft4k7k_istft", "stft4k10k_stft_dif", "welch1k3k_y", "welch1k3k_x", "welch8k3k_y", "welch8k3k_x", "welch16k3k_y", "welch16k3k_x", "welchl4k_yc", "welchl4k_xc", "stftl10k_stft_cls",
    "fir16k5k_firILb1E", "fir4k5k_firILi1E", "fir4k5k_firILi2E", "fir4k6k_fir3ILi0E", "deconv8k10k_deconv_p", "k_csm_gemm64",
]


# kernels whose workgroups wait for each other inside one launch (whole grid resident at once)
RESIDENT_GRID = []  # (none since round 4: the one-launch Welch kernel lives in tools/exp only)

HASH_PATH = os.path.join(LIB_DIR, "libdsptoolbox_amd.srchash")


def _source_hash() -> str:
    """sha256 over the contents of every source the library is built from (names included).
    Content, not modification times: a checkout or a copy to another machine touches every file
    without changing anything, and must not trigger a two-minute rebuild."""
    import hashlib
    h = hashlib.sha256()
    files = []
    for root in (CSRC, os.path.join(HERE, "..", "include")):
        for dp, _, names in os.walk(root):
            for f in names:
                if f.endswith((".hip", ".hpp", ".h")):
                    files.append(os.path.join(dp, f))
    for path in sorted(files, key=lambda q: os.path.relpath(q, HERE)):
        h.update(os.path.relpath(path, HERE).encode())
        with open(path, "rb") as fh:
            h.update(fh.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def _up_to_date() -> bool:
    if not (os.path.exists(LIB_PATH) and os.path.exists(HASH_PATH)):
        return False
    with open(HASH_PATH) as fh:
        return fh.read().strip() == _source_hash()


def _parse_resources(text: str) -> dict:
    """{mangled kernel name: {vgpr, agpr, sgpr, scratch, lds, occupancy}} from the compiler's
    -Rpass-analysis=kernel-resource-usage remarks."""
    import re
    out, cur = {}, None
    keys = {"VGPRs": "vgpr", "AGPRs": "agpr", "TotalSGPRs": "sgpr", "ScratchSize [bytes/lane]": "scratch",
            "LDS Size [bytes/block]": "lds", "Occupancy [waves/SIMD]": "occupancy", "VGPRs Spill": "vgpr_spill"}
    for line in text.splitlines():
        m = re.search(r"remark: (?:.*: )?Function Name: (\S+)", line)
        if m:
            cur = out.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark:\s+(?:.*?:\s+)?([A-Za-z][A-Za-z \[\]/]+): (\d+)", line)
        if m and cur is not None and m.group(1).strip() in keys:
            cur[keys[m.group(1).strip()]] = int(m.group(2))
    return out


def _without_remarks(text: str) -> str:
    keep, skip = [], 0
    for line in text.splitlines(True):
        if "-Rpass-analysis=kernel-resource-usage" in line:
            skip = 2  # the remark is followed by a source line and a caret line
            continue
        if skip and (line.lstrip().startswith("|") or line.strip().startswith("^") or "|" in line[:8]):
            skip -= 1
            continue
        skip = 0
        keep.append(line)
    return "".join(keep)


def kernel_fingerprints(lib_path: str = LIB_PATH) -> dict:
    """{mangled kernel name: first 16 hex digits of the sha256 of its machine code} from the gfx950 code object inside
    the library (the uncompressed clang offload bundle in .hip_fatbin; plain ELF64 parsing, no tools).  What a profile
    file records beside its counters, so that a reader can tell whether they were taken on the kernel that runs now
    (bench.py: roofline.traffic_kernel_current)."""
    import hashlib
    import struct
    blob = open(lib_path, "rb").read()
    start = blob.find(b"__CLANG_OFFLOAD_BUNDLE__")
    if start < 0:
        return {}
    n_entries = struct.unpack_from("<Q", blob, start + 24)[0]
    pos, elf = start + 32, None
    for _ in range(n_entries):
        off, size, id_len = struct.unpack_from("<QQQ", blob, pos)
        ident = blob[pos + 24:pos + 24 + id_len]
        pos += 24 + id_len
        if b"gfx950" in ident and size:
            elf = blob[start + off:start + off + size]
    if elf is None or elf[:4] != b"\x7fELF":
        return {}
    shoff, = struct.unpack_from("<Q", elf, 0x28)
    shentsize, shnum, _ = struct.unpack_from("<HHH", elf, 0x3A)
    secs = [struct.unpack_from("<IIQQQQIIQQ", elf, shoff + i * shentsize) for i in range(shnum)]
    out = {}
    for name, typ, flags, addr, offset, size, link, info, align, entsize in secs:
        if typ != 2:  # SHT_SYMTAB
            continue
        str_off = secs[link][4]
        for i in range(size // 24):
            st_name, st_info, _, st_shndx, st_value, st_size = struct.unpack_from("<IBBHQQ", elf, offset + 24 * i)
            if (st_info & 0xF) != 2 or st_size == 0 or st_shndx == 0 or st_shndx >= shnum:  # STT_FUNC, defined
                continue
            sec = secs[st_shndx]
            code = elf[sec[4] + st_value - sec[3]:sec[4] + st_value - sec[3] + st_size]
            end = elf.index(b"\0", str_off + st_name)
            out[elf[str_off + st_name:end].decode()] = hashlib.sha256(code).hexdigest()[:16]
    return out


INFO_PATH = os.path.join(LIB_DIR, "build_info.json")


def _run_compiler_id() -> str:
    try:
        out = subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--version"], stdout=subprocess.PIPE,
                             stderr=subprocess.STDOUT, text=True, timeout=60).stdout.splitlines()
    except (OSError, subprocess.SubprocessError):
        return ""
    return " | ".join(l.strip() for l in out[:2])


def _run_demangle(fp: dict) -> dict:
    names = sorted(fp)
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), stdout=subprocess.PIPE, stderr=subprocess.DEVNULL,
                             text=True, check=True).stdout.splitlines()
    except (OSError, subprocess.CalledProcessError):
        return {}
    return {d: fp[m] for m, d in zip(names, out)} if len(out) == len(names) else {}


def write_build_info() -> dict:
    """lib/build_info.json: the compiler (first two lines of `hipcc --version`), the source hash of the library beside it and
    {demangled kernel name: fingerprint}.  Written where the library is BUILT (build_library after a compile,
    __graft_entry__.build() when the file is missing or belongs to another library) -- it runs hipcc and c++filt, and a
    process that has initialised the GPU (bench.py; anything under rocprofv3) must not start programs: hipcc --version alone
    runs rocm_agent_enumerator, a python script, and a GPU box refuses that exec.  Readers use build_info()."""
    import json
    info = dict(source_hash=open(HASH_PATH).read().strip() if os.path.exists(HASH_PATH) else "",
                compiler=_run_compiler_id(), fingerprints=_run_demangle(kernel_fingerprints()))
    with open(INFO_PATH, "w") as fh:
        json.dump(info, fh, indent=1, sort_keys=True)
    return info


def build_info() -> dict:
    """The record write_build_info left for the library in the tree ({} if there is none or it is another library's).
    No program is started."""
    import json
    try:
        with open(INFO_PATH) as fh:
            info = json.load(fh)
        with open(HASH_PATH) as fh:
            return info if info.get("source_hash") == fh.read().strip() else {}
    except (OSError, ValueError):
        return {}


def compiler_id() -> str:
    """The compiler of the library in the tree (kernel fingerprints are only comparable between libraries built by the same
    one); "" when not recorded."""
    return build_info().get("compiler", "")


def demangled_fingerprints() -> dict:
    """{demangled kernel name as rocprofv3 prints it: fingerprint} of the library in the tree; {} when not recorded."""
    return build_info().get("fingerprints", {})


LAST_BUILD = None  # "compiled" or "reused" (the library's source hash matched): what build_library last did


def build_library(force: bool = False, verbose: bool = True) -> str:
    """hipcc --offload-arch=gfx950 -> dsptoolbox_amd/lib/libdsptoolbox_amd.so

    Safe to call from several ranks at once: one process compiles (file lock) into a
    temporary name and renames it into place, the others wait and reuse the result."""
    global LAST_BUILD
    os.makedirs(LIB_DIR, exist_ok=True)
    if not force and _up_to_date():
        LAST_BUILD = "reused"
        return LIB_PATH
    import fcntl
    with open(os.path.join(LIB_DIR, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and _up_to_date():  # another rank built it while we waited
                LAST_BUILD = "reused"
                return LIB_PATH
            hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
            tmp = LIB_PATH + f".tmp{os.getpid()}"
            # the hash of what is about to be compiled: taken now, not after the two minutes hipcc runs (a source
            # edited meanwhile would otherwise be recorded as built)
            hash_at_start = _source_hash()
            cmd = [hipcc] + FLAGS + REMARK_FLAGS + ["-o", tmp] + SOURCES + ["-ldl", "-pthread"]
            if verbose:
                print(" ".join(cmd), file=sys.stderr, flush=True)
            try:
                proc = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
                res = _parse_resources(proc.stderr)
                if proc.returncode != 0:
                    sys.stderr.write(_without_remarks(proc.stderr))
                    raise subprocess.CalledProcessError(proc.returncode, cmd)
                # the gate must not fail open: no parsed remarks, or a listed kernel that matches
                # nothing (a rename, a changed remark format), is an error of the build itself
                if not res:
                    raise RuntimeError("no kernel-resource-usage remarks parsed from the compiler output; "
                                       "the no-scratch gate cannot be checked")
                unmatched = [t for t in NO_SCRATCH if not any(t in k for k in res)]
                if unmatched:
                    raise RuntimeError("NO_SCRATCH entries that match no compiled kernel: %r" % unmatched)
                bad = {k: v["scratch"] for k, v in res.items()
                       if v.get("scratch", 0) > 0 and any(t in k for t in NO_SCRATCH)}
                if bad and not os.environ.get("DSPTOOLBOX_AMD_ALLOW_SCRATCH"):
                    raise RuntimeError("hot kernels compiled with scratch (register spills), bytes/lane: %r" % bad)
                # kernels that wait for other workgroups of their own grid: the occupancy query is only
                # right up to 80 SGPRs (MI355X_MICROARCH.md, Residency) -- keep them there
                wide = {k: v.get("sgpr", 0) for k, v in res.items()
                        if any(t in k for t in RESIDENT_GRID) and v.get("sgpr", 0) > 80}
                if wide:
                    raise RuntimeError("resident-grid kernels with more than 80 SGPRs: %r" % wide)
                rest = _without_remarks(proc.stderr)
                if rest.strip() and verbose:
                    sys.stderr.write(rest)  # warnings of a successful build are not swallowed
                import json
                with open(RES_PATH, "w") as fh:
                    json.dump(res, fh, indent=1, sort_keys=True)
                os.replace(tmp, LIB_PATH)
                with open(HASH_PATH, "w") as fh:
                    fh.write(hash_at_start + "\n")
                write_build_info()
                LAST_BUILD = "compiled"
            finally:
                if os.path.exists(tmp):
                    os.remove(tmp)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB_PATH


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv))
