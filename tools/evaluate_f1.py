#!/usr/bin/env python3
"""Downstream check of the embeddings, following the reference README's protocol (README.md:51-73):
multi-class logistic regression on Z, train ratio 10 %...90 %, several random splits, micro / macro F1.

    python tools/evaluate_f1.py --embeddings out/Z.npy --labels data_root/Y [--baseline data_root/C.npy]

`Y` holds one `id<TAB>class` per line in the order of `V` (the reference ships such a file for the karate
graph, tests/data_root/Y, but no code reads it).  Not on the hot path; needs scikit-learn.
"""
import argparse
from pathlib import Path

import numpy as np


def f1_table(Z: np.ndarray, y: np.ndarray, ratios, runs: int, seed: int = 0):
    from sklearn.linear_model import LogisticRegression
    from sklearn.metrics import f1_score
    from sklearn.model_selection import train_test_split
    rows = []
    for ratio in ratios:
        micro, macro = [], []
        for run in range(runs):
            tr, te = train_test_split(np.arange(len(y)), train_size=ratio, random_state=seed + run, stratify=None)
            if len(np.unique(y[tr])) < 2:
                continue
            clf = LogisticRegression(max_iter=2000).fit(Z[tr], y[tr])
            pred = clf.predict(Z[te])
            micro.append(f1_score(y[te], pred, average="micro"))
            macro.append(f1_score(y[te], pred, average="macro"))
        rows.append((ratio, float(np.mean(micro)), float(np.mean(macro))))
    return rows


def read_labels(path: Path) -> np.ndarray:
    labels = [line.split("\t")[1] for line in Path(path).read_text().strip().split("\n")]
    classes = {c: i for i, c in enumerate(sorted(set(labels)))}
    return np.array([classes[c] for c in labels])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--embeddings", type=Path, required=True)
    ap.add_argument("--labels", type=Path, required=True)
    ap.add_argument("--baseline", type=Path, default=None, help="content embeddings (C.npy) to compare against")
    ap.add_argument("--runs", type=int, default=10)
    args = ap.parse_args()
    y = read_labels(args.labels)
    ratios = [r / 10 for r in range(1, 10)]
    tables = {"embeddings": f1_table(np.load(args.embeddings), y, ratios, args.runs)}
    if args.baseline:
        tables["baseline"] = f1_table(np.load(args.baseline), y, ratios, args.runs)
    for name, rows in tables.items():
        print(f"{name}\n  train%  " + " ".join(f"{int(r * 100):5d}" for r, _, _ in rows))
        print("  microF1 " + " ".join(f"{m:5.3f}" for _, m, _ in rows))
        print("  macroF1 " + " ".join(f"{m:5.3f}" for _, _, m in rows))


if __name__ == "__main__":
    main()
