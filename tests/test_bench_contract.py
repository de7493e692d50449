"""bench.py contract checks that need no GPU: argument defaults and the JSON line's shape."""
import ast
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_defaults_and_keys():
    src = open(os.path.join(ROOT, "bench.py")).read()
    tree = ast.parse(src)
    # defaults: N=1, steps/warmup present
    assert '"--gpus", type=int, default=1' in src
    assert '"--steps", type=int' in src and '"--warmup", type=int' in src
    for key in ('"metric"', '"value"', '"unit"', '"n_gpus"', '"steps"', '"warmup"', '"ms_per_step"',
                '"higher_is_better"', '"scaling"', '"vs_baseline"', '"dtype"', '"data"', '"config"',
                '"roofline"', '"cpu_baseline"'):
        assert key in src, key
    # exactly one print to stdout (the JSON line); everything else goes to stderr
    prints = [n for n in ast.walk(tree) if isinstance(n, ast.Call) and getattr(n.func, "id", "") == "print"]
    to_stdout = [p for p in prints if not any(k.arg == "file" for k in p.keywords)]
    assert len(to_stdout) == 1
    # only the cpu_baseline leg touches the oracle
    fn = {n.name: n for n in tree.body if isinstance(n, ast.FunctionDef)}
    for name, node in fn.items():
        uses_oracle = "oracle" in ast.get_source_segment(src, node)
        assert uses_oracle == (name == "cpu_baseline"), name


def test_product_never_imports_oracle():
    for base, _, files in os.walk(os.path.join(ROOT, "centerpoly_amd")):
        for f in files:
            if f.endswith(".py"):
                text = open(os.path.join(base, f)).read()
                assert "import oracle" not in text and "from oracle" not in text, f
    for f in ("main.py", "test.py"):
        text = open(os.path.join(ROOT, f)).read()
        assert "oracle" not in text, f
