"""Fixture for the context construction of the text feature extractor: runs the REAL reference function
(/root/reference/src/feature_extractors/text/utils.py:61-92, get_utterance_with_context) on a seeded MELD-shaped table.
The module itself cannot be imported here (it needs `munch`, which is not installed), so the ONE function is taken out of the
file's syntax tree and executed as it stands - nothing of it is stored: the fixture holds the table and the strings it returned.
Run in the build container only:  python tests/golden/make_golden_context.py"""
import ast
import json
import os

import numpy as np
import pandas as pd

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = "/root/reference/src/feature_extractors/text/utils.py"


def reference_function():
    tree = ast.parse(open(SRC).read())
    fn = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "get_utterance_with_context")
    ns = {}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), SRC, "exec"), ns)
    return ns["get_utterance_with_context"]


def table(seed=5):
    g = np.random.default_rng(seed)
    words = ["oh", "my", "god", "what", "no", "you", "really", "fine", "okay", "…", "it’s", "a", "trap", "!", "?"]
    rows = []
    for d in (3, 0, 7, 12, 5):                              # dialogue ids in order of appearance, not sorted
        n = int(g.integers(1, 7))
        ids = sorted(g.choice(12, size=n, replace=False).tolist())      # gaps in the utterance ids (the corrupted-clip rows are dropped)
        for u in ids:
            k = int(g.integers(1, 6))
            rows.append((" ".join(words[int(x)] for x in g.integers(0, len(words), size=k)), int(d), int(u)))
    rows.append(("alone", 99, 4))                           # a one-utterance dialogue
    order = g.permutation(len(rows))                         # rows of a dialogue are not contiguous / not in id order
    return pd.DataFrame([rows[i] for i in order], columns=["Utterance", "Dialogue_ID", "Utterance_ID"])


if __name__ == "__main__":
    fn = reference_function()
    df = table()
    out = {"separator": "</s>", "utterances": df["Utterance"].tolist(), "dialogue_ids": df["Dialogue_ID"].tolist(),
           "utterance_ids": df["Utterance_ID"].tolist(), "contexts": [fn(df, i, "</s>") for i in range(len(df))]}
    with open(os.path.join(HERE, "text_contexts.json"), "w", encoding="utf-8") as f:
        json.dump(out, f, ensure_ascii=False, indent=0)
    print(len(df), "rows;", out["contexts"][0], "|", out["contexts"][-1])
