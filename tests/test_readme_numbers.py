"""README.md quotes measured figures only through tools/readme_numbers.py (a rounded-up "8.0" had appeared next to a measured
7.89): the block between the markers must be exactly what the tool generates from the bench record it names, truncated, and that record must
be of the newest round on disk -- the driver's BENCH_rNN.json or the builder's profiles/rNN_bench.json."""
import importlib.util
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def tool():
    spec = importlib.util.spec_from_file_location("rc_readme_numbers", os.path.join(ROOT, "tools", "readme_numbers.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_readme_block_is_generated_from_the_newest_bench_record():
    t = tool()
    text = open(os.path.join(ROOT, "README.md")).read()
    assert t.BEGIN in text and t.END in text
    block = text[text.index(t.BEGIN):text.index(t.END) + len(t.END)]
    named = re.search(r"read from `([^`]+)`", block).group(1)
    src = {rel: (rnd, b) for rnd, _, rel, b in t.sources()}
    assert named in src, f"README names {named}, which is not a bench record on disk"
    newest = max(r for r, _ in src.values())
    assert src[named][0] == newest, f"README quotes round {src[named][0]}, round {newest} is on disk: python3 tools/readme_numbers.py --write"
    assert block == t.block(named, src[named][1]), "README's figures differ from its source: python3 tools/readme_numbers.py --write"
    # no measured Grays/s figure outside the block
    outside = text.replace(block, "")
    assert not re.search(r"\d\.\d+\s*Grays/s", outside), "a Grays/s figure outside the generated block"


def test_figures_are_truncated_not_rounded():
    t = tool()
    assert t.g(7886.4) == "7.88" and t.g(7999.9) == "7.99" and t.g(8000.0) == "8.00"
