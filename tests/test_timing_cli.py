"""tools/time_tokenizer.py -- counterpart of the reference's scripts/timing/time_tokenizer.py."""
import csv
import gzip
import importlib.util
import io
import json
import os

import pytest

from conftest import ROOT

SAMPLES = ["This is a #test! Testing, Testing, 1 2 3", "see http://a.b/c or mail me@x.org", "camelCase 日本語 🤓",
           "quote \" comma , tab\t", "", "  padded  "]


def _cli():
    spec = importlib.util.spec_from_file_location("time_tokenizer", os.path.join(ROOT, "tools", "time_tokenizer.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _write(tmp_path, fmt, gz=False):
    name = {"csv-json": "in.csv", "lines": "in.txt", "jsonl": "in.jsonl"}[fmt] + (".gz" if gz else "")
    path = str(tmp_path / name)
    buf = io.StringIO()
    if fmt == "csv-json":   # the reference's input: csv rows, second column = JSON-encoded string
        w = csv.writer(buf)
        for i, s in enumerate(SAMPLES):
            w.writerow([i, json.dumps(s)])
    elif fmt == "lines":
        buf.write("\n".join(s.replace("\t", " ") for s in SAMPLES) + "\n")
    else:
        for i, s in enumerate(SAMPLES):
            buf.write(json.dumps(s if i % 2 else {"text": s, "id": i}) + "\n")
    with (gzip.open(path, "wt", encoding="utf-8", newline="") if gz else open(path, "w", encoding="utf-8", newline="")) as f:
        f.write(buf.getvalue())
    return path


@pytest.mark.parametrize("fmt,gz", [("csv-json", False), ("csv-json", True), ("lines", False), ("jsonl", True)])
def test_reader_formats(tmp_path, fmt, gz):
    cli = _cli()
    got = list(cli.read_texts(_write(tmp_path, fmt, gz), fmt))
    want = [s.strip() for s in SAMPLES]
    if fmt == "lines":
        want = [s.replace("\t", " ").strip() for s in SAMPLES]
    assert got == want
    assert [len(b) for b in cli.batches(iter(range(10)), 4)] == [4, 4, 2]


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["tokens", "split", "features", "matrix"])
def test_cli_runs_and_writes_reference_tokens(tmp_path, gpu, oracle, mode):
    cli = _cli()
    path = _write(tmp_path, "csv-json")
    out = str(tmp_path / "out.tsv") if mode in ("tokens", "features") else None
    with open(os.devnull, "w") as log:
        s = cli.run(path, mode, "csv-json", batch_size=4, mincount=1, outfile=out, log=log)
    assert s["lines"] == len(SAMPLES) and s["chars"] == sum(len(x.strip()) for x in SAMPLES)
    if out:
        lines = open(out, encoding="utf-8").read().split("\n")[:-1]
        want = ["\t".join(oracle.tokenize(x.strip()) if x.strip() else []) for x in SAMPLES]
        assert lines == want
