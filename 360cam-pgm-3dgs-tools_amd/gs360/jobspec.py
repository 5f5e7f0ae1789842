"""Parse the ffmpeg-shaped job argv produced by the planner back into a structured job.

Why parse instead of passing a private struct: the unmodified GUI rewrites each argv between planning and
execution (reference gs360_GUI.py:19081-19148 -- it moves -ss/-to, drops fps=, prepends select='eq(n\\,i)+...',
inserts -frame_pts/-copyts), and calls run_one(argv) (gs360_GUI.py:19299).  The argv therefore IS the job
description at the drop-in seam.
"""
import pathlib
from dataclasses import dataclass, field
from typing import Dict, List, Optional

IMAGE_EXTS = {".tif", ".tiff", ".jpg", ".jpeg", ".png"}


class JobParseError(ValueError):
    pass


@dataclass
class JobSpec:
    program: str
    src: pathlib.Path
    dst: pathlib.Path
    v360: Dict[str, str]
    filters: List[str] = field(default_factory=list)     # the other filters of the -vf chain, in order
    options: Dict[str, str] = field(default_factory=dict)  # remaining "-key value" pairs
    flags: List[str] = field(default_factory=list)
    filters_after: List[str] = field(default_factory=list)    # the members of `filters` that follow v360 in the chain
    input_options: List[str] = field(default_factory=list)    # "-key", "value" tokens that precede -i (order kept)
    output_options: List[str] = field(default_factory=list)   # "-key", "value" tokens that follow -i (except -vf)

    # ---- v360 parameters ------------------------------------------------------------------------
    @property
    def output_projection(self) -> str:
        return self.v360.get("output", "")

    @property
    def input_projection(self) -> str:
        return self.v360.get("input", "")

    def fnum(self, key: str, default: Optional[float] = None) -> float:
        if key not in self.v360:
            if default is None:
                raise JobParseError(f"v360 parameter '{key}' missing")
            return default
        try:
            return float(self.v360[key])
        except ValueError as exc:
            raise JobParseError(f"v360 parameter {key}={self.v360[key]!r} is not a number") from exc

    @property
    def width(self) -> int:
        return int(self.fnum("w"))

    @property
    def height(self) -> int:
        return int(self.fnum("h"))

    @property
    def interp(self) -> str:
        return self.v360.get("interp", "linear")

    # ---- classification -------------------------------------------------------------------------
    @property
    def is_still_image(self) -> bool:
        return self.src.suffix.lower() in IMAGE_EXTS

    @property
    def jpeg_q(self) -> Optional[int]:
        q = self.options.get("-q:v")
        return int(q) if q and q.isdigit() else None

    def filter_named(self, name: str) -> Optional[str]:
        for f in self.filters:
            if f.split("=", 1)[0] == name:
                return f
        return None


def split_filter_chain(chain: str) -> List[str]:
    """Split an ffmpeg -vf chain on top-level commas (commas inside '...' or escaped as \\, stay)."""
    parts, cur, quoted, i = [], [], False, 0
    while i < len(chain):
        ch = chain[i]
        if ch == "\\" and i + 1 < len(chain):
            cur.append(chain[i:i + 2])
            i += 2
            continue
        if ch == "'":
            quoted = not quoted
        if ch == "," and not quoted:
            parts.append("".join(cur))
            cur = []
        else:
            cur.append(ch)
        i += 1
    if cur:
        parts.append("".join(cur))
    return [p for p in (s.strip() for s in parts) if p]


_VALUELESS = {"-hide_banner", "-y", "-n", "-nostdin", "-copyts", "-an", "-sn"}


def parse_job_argv(argv: List[str]) -> JobSpec:
    if not argv or len(argv) < 4:
        raise JobParseError("job argv too short")
    program, dst = argv[0], argv[-1]
    body = argv[1:-1]
    src = None
    chain = None
    options: Dict[str, str] = {}
    flags: List[str] = []
    before_i: List[str] = []
    after_i: List[str] = []
    i = 0
    while i < len(body):
        tok = body[i]
        if tok in _VALUELESS:
            flags.append(tok)
            i += 1
            continue
        if not tok.startswith("-") or i + 1 >= len(body):
            raise JobParseError(f"unexpected token {tok!r} in job argv")
        val = body[i + 1]
        if tok == "-i":
            src = val
        elif tok in ("-vf", "-filter:v"):
            chain = val
        else:
            options[tok] = val
            (before_i if src is None else after_i).extend([tok, val])
        i += 2
    if src is None:
        raise JobParseError("job argv has no -i <input>")
    if chain is None:
        raise JobParseError("job argv has no -vf filter chain")
    v360: Optional[Dict[str, str]] = None
    others: List[str] = []
    after: List[str] = []
    for flt in split_filter_chain(chain):
        if flt.startswith("v360="):
            if v360 is not None:
                raise JobParseError("more than one v360 filter in the chain")
            v360 = {}
            for kv in flt[len("v360="):].split(":"):
                if "=" not in kv:
                    raise JobParseError(f"malformed v360 option {kv!r}")
                k, v = kv.split("=", 1)
                v360[k] = v
        else:
            others.append(flt)
            if v360 is not None:
                after.append(flt)
    if v360 is None:
        raise JobParseError("filter chain has no v360 filter")
    return JobSpec(program, pathlib.Path(src), pathlib.Path(dst), v360, others, options, flags, after, before_i, after_i)
