"""the RGCN_* environment switches are a product surface.  Every switch the sources read must be listed in
DESIGN.md section 7 (name, default, meaning), nothing listed there may be dead, and the ones that change SEMANTICS (not just the
kernel that runs) must be named as such."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
READ = re.compile(r'(?:os\.environ\.get\(\s*|os\.environ\[\s*|getenv\(\s*|setdefault\(\s*)"(RGCN_[A-Z0-9_]+)"|"(RGCN_[A-Z0-9_]+)"\s+(?:not\s+)?in\s+os\.environ')
SEMANTIC = {"RGCN_DEFERRED_CHECKS", "RGCN_DETERMINISTIC", "RGCN_SYNTHETIC"}       # change what is computed / when errors surface
TIMING_ONLY = {"RGCN_BWD_ABL", "RGCN_RANK_ABLATE"}                                # wrong results by design (tools)


def _sources():
    for base in ("torch-rgcn_amd", "bench.py", "__graft_entry__.py"):
        path = os.path.join(ROOT, base)
        if os.path.isfile(path):
            yield path
            continue
        for d, _, files in os.walk(path):
            for f in files:
                if f.endswith((".py", ".hip", ".cpp", ".h")):
                    yield os.path.join(d, f)


def _switches_read():
    found = {}
    for path in _sources():
        for m in READ.finditer(open(path, encoding="utf-8", errors="replace").read()):
            found.setdefault(m.group(1) or m.group(2), set()).add(os.path.relpath(path, ROOT))
    return found


def _design_section7():
    text = open(os.path.join(ROOT, "DESIGN.md"), encoding="utf-8").read()
    a = text.index("## 7. Environment switches")
    return text[a:text.index("\n## 8.", a)]


def test_every_switch_read_by_the_sources_is_documented():
    sec = _design_section7()
    missing = {k: sorted(v) for k, v in _switches_read().items() if f"`{k}`" not in sec and f"`{k}=" not in sec}
    assert not missing, f"undocumented environment switches (add them to DESIGN.md section 7): {missing}"


def test_no_documented_switch_is_dead():
    read = set(_switches_read())
    listed = set(re.findall(r"`(RGCN_[A-Z0-9_]+)(?:=[^`]*)?`", _design_section7()))
    dead = sorted(listed - read)
    assert not dead, f"DESIGN.md section 7 lists switches nothing reads any more: {dead}"


def test_semantic_switches_are_flagged():
    sec = _design_section7()
    a = sec.index("change semantics")
    para = sec[a:a + 1200]
    for k in SEMANTIC | TIMING_ONLY:
        assert k in para, f"{k} changes results or error behaviour and must be named in the 'change semantics' paragraph"
