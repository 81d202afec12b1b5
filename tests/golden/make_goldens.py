#!/usr/bin/env python3
"""Golden-vector generator (runs ONLY in the build container, where /root/reference exists).

The reference ships no tests and no fixtures for the retrieval hot path, and its
modules cannot be imported here (duckdb / spaCy / sentence-transformers are absent; one f-string in
indexer/bm25_indexer.py needs Python >= 3.12).  What *can* be done is to run the reference's own
function bodies: this script reads the reference files as text, pulls individual function / class
definitions out of their AST **at run time**, and executes them against in-memory stand-ins for the
three things that are not available (the DuckDB connection, the sentence encoder, the FastAPI app).
Nothing from the reference is stored in this repository: the outputs written next to this file are
data only (inputs + the values the reference functions returned).

Stand-ins and what they assume (these are the places where behaviour lives in a third-party engine
and is therefore *not* pinned by executing the reference):
  * DuckDB `REAL` columns (idf_score, stat_value) round to float32         bm25_indexer.py:110,118
  * DuckDB `LOG` is log10 (used only to build synthetic idf columns)       bm25_indexer.py:138
  * `ORDER BY tf.doc_id` / `SELECT DISTINCT` on the candidate query        bm25_indexer.py:436-446
  * `ROW_NUMBER() OVER(PARTITION BY doc_id)` has no ORDER BY; the stand-in numbers a document's
    chunks by ascending chunk_id (the order they were inserted)            reranker_api.py:49-58
  * FLOAT[768] cells arrive in pandas as float32 ndarrays                  reranker_api.py:61
  * rows of the SQL result are ordered by (doc_id, chunk_id)

Usage:  python tests/golden/make_goldens.py            (rewrites tests/golden/*.json|*.npz)
"""
import ast
import asyncio
import json
import logging
import math
import os
import sys
import types
import warnings
from collections import defaultdict
from typing import Dict, List, Optional, Tuple, Union

import numpy as np

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


# --------------------------------------------------------------------------------------------
# AST helpers
# --------------------------------------------------------------------------------------------
def _read(rel):
    with open(os.path.join(REF, rel), encoding="utf-8") as f:
        return f.read()


def _pick(tree, names, strip_decorators=True):
    """Return a Module holding only the named top-level defs of `tree` (order preserved)."""
    body = []
    for node in tree.body:
        if isinstance(node, (ast.FunctionDef, ast.AsyncFunctionDef, ast.ClassDef)) and node.name in names:
            if strip_decorators and not isinstance(node, ast.ClassDef):
                node.decorator_list = []
            body.append(node)
    missing = set(names) - {n.name for n in body}
    if missing:
        raise RuntimeError(f"reference definitions not found: {sorted(missing)}")
    mod = ast.Module(body=body, type_ignores=[])
    ast.fix_missing_locations(mod)
    return mod


def _exec(mod, ns, filename):
    exec(compile(mod, filename, "exec"), ns)
    return ns


# --------------------------------------------------------------------------------------------
# reranker_api.py functions
# --------------------------------------------------------------------------------------------
def load_reranker_namespace():
    import pandas as pd
    import yaml
    from pydantic import BaseModel
    from sklearn.metrics.pairwise import cosine_similarity
    from urllib.parse import urlparse

    tree = ast.parse(_read("reranker/reranker_api.py"))
    with open(os.path.join(REF, "reranker/config.yaml"), encoding="utf-8") as f:
        config = yaml.safe_load(f)

    class HTTPException(Exception):
        def __init__(self, status_code, detail=None):
            super().__init__(f"{status_code}: {detail}")
            self.status_code = status_code
            self.detail = detail

    ns = {
        "np": np, "pd": pd, "cosine_similarity": cosine_similarity, "urlparse": urlparse,
        "List": List, "Dict": Dict, "Tuple": Tuple, "Optional": Optional, "Union": Union,
        "BaseModel": BaseModel, "HTTPException": HTTPException, "config": config,
        "logger": logging.getLogger("ref-reranker"),
    }
    names = ["RerankRequest", "WindowScore", "DocumentScore", "RerankResponse", "extract_domain",
             "apply_domain_cap", "hybrid_diversification", "create_sliding_windows",
             "calculate_similarity", "get_new_similarity", "normalise_similarities",
             "apply_positional_weighting", "rerank"]
    _exec(_pick(tree, names), ns, "<reference reranker_api.py>")
    ns["__tree__"] = tree
    return ns


class StubRerankDB:
    """In-memory stand-in for reranker_api.Database.get_documents_by_ids (reranker_api.py:27-63)."""

    def __init__(self, urls, chunks, emb):
        # urls: list of (id:int, url, title, text); chunks: list of (chunk_id:int, doc_id:int);
        # emb: dict chunk_id -> float32[768]
        self.urls, self.chunks, self.emb = urls, chunks, emb

    def get_documents_by_ids(self, doc_ids):
        import pandas as pd
        if isinstance(doc_ids, str):
            doc_ids = [doc_ids]
        want = {int(x) for x in doc_ids}
        groups = {}
        for (i, url, title, text) in sorted(self.urls):          # FIRST(): lowest id of the group
            if i not in want:
                continue
            key = url[:url.index("?")] if "?" in url else url
            groups.setdefault(key, (i, title, url, text))        # MIN(id) == first, rows sorted by id
        keep = {g[0]: g for g in groups.values()}
        rn = defaultdict(int)
        rows = []
        for (cid, did) in sorted(self.chunks, key=lambda r: (r[1], r[0])):
            rn[did] += 1
            if did in keep and rn[did] <= 10:
                g = keep[did]
                rows.append({"id": str(g[0]), "title": g[1], "url": g[2], "text": g[3],
                             "chunk_id": cid, "doc_id": did, "chunk_text": f"chunk {cid}",
                             "rn": rn[did], "embedding": self.emb[cid].astype(np.float32)})
        cols = ["id", "title", "url", "text", "chunk_id", "doc_id", "chunk_text", "rn", "embedding"]
        return pd.DataFrame(rows, columns=cols)


class StubEncoder:
    def __init__(self, vec):
        self.vec = vec

    def encode(self, text):
        return self.vec


def run_rerank(ns, db, qvec, doc_ids, sims, diversification=True):
    """Execute the reference's `rerank` coroutine body verbatim against the stand-ins."""
    ns["database"] = db
    ns["embedding_model"] = StubEncoder(qvec)
    ns["config"]["similarity"]["diversification"] = diversification
    req = ns["RerankRequest"](doc_ids=[str(d) for d in doc_ids], similarities=sims, query="q")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return asyncio.run(ns["rerank"](req))


def run_rerank_stages(ns, db, qvec, doc_ids, sims):
    """Run the plain assignment statements of `rerank` (up to the per-document pooling) one by one
    and snapshot the chunk table after each stage."""
    tree = ns["__tree__"]
    fn = next(n for n in tree.body if isinstance(n, ast.AsyncFunctionDef) and n.name == "rerank")
    try_node = next(n for n in fn.body if isinstance(n, ast.Try))
    scope = dict(ns)
    scope["database"] = db
    scope["embedding_model"] = StubEncoder(qvec)
    scope["request"] = ns["RerankRequest"](doc_ids=[str(d) for d in doc_ids], similarities=sims, query="q")
    snaps = []
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for st in try_node.body:
            if not isinstance(st, ast.Assign):
                continue
            tgt = ast.unparse(st.targets[0])
            if tgt == "document_scores":
                break
            mod = ast.Module(body=[st], type_ignores=[])
            exec(compile(mod, "<reference rerank stmt>", "exec"), scope)
            if tgt.startswith("documents") or tgt == "reranked_documents":
                df = scope["reranked_documents"] if tgt == "reranked_documents" else scope["documents"]
                if "new_similarity" in df.columns:
                    snaps.append({
                        "target": tgt,
                        "doc_id": [int(x) for x in df["doc_id"]],
                        "chunk_id": [int(x) for x in df["chunk_id"]],
                        "new_similarity": [float(x) for x in df["new_similarity"]],
                        "old_similarity": [float(x) for x in df["old_similarity"]],
                    })
    return snaps


# --------------------------------------------------------------------------------------------
# bm25_indexer.py : BM25.search, BM25._get_corpus_stats
# --------------------------------------------------------------------------------------------
def load_bm25_methods():
    src = _read("indexer/bm25_indexer.py")
    lines = src.split("\n")
    # indexer/bm25_indexer.py:508 nests single quotes inside a single-quoted f-string, which only
    # parses on Python >= 3.12.  Re-quote the inner literal (same meaning) so 3.10 can parse it.
    bad = [i for i, l in enumerate(lines) if "f'{title or 'N/A'}" in l]
    if len(bad) != 1:
        raise RuntimeError("expected exactly one 3.12-only f-string in bm25_indexer.py")
    lines[bad[0]] = lines[bad[0]].replace("'N/A'", '"N/A"')
    tree = ast.parse("\n".join(lines))
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "BM25")
    methods = [n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name in ("search", "_get_corpus_stats")]
    mod = ast.Module(body=methods, type_ignores=[])
    ast.fix_missing_locations(mod)
    ns = {"defaultdict": defaultdict, "List": List, "Dict": Dict, "Tuple": Tuple, "Optional": Optional,
          "math": math, "logging": logging}
    _exec(mod, ns, "<reference bm25_indexer.py>")
    return ns["search"], ns["_get_corpus_stats"]


def load_bm25_build_method():
    """BM25._process_document_batch (bm25_indexer.py:196-243): the per-batch table builder of build_index, sequential
    branch (< 50 documents).  Same AST route as `search`; the tokeniser is the only stand-in (spaCy is absent)."""
    src = _read("indexer/bm25_indexer.py")
    lines = src.split("\n")
    bad = [i for i, l in enumerate(lines) if "f'{title or 'N/A'}" in l]
    if len(bad) != 1:
        raise RuntimeError("expected exactly one 3.12-only f-string in bm25_indexer.py")
    lines[bad[0]] = lines[bad[0]].replace("'N/A'", '"N/A"')
    tree = ast.parse("\n".join(lines))
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "BM25")
    methods = [n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name == "_process_document_batch"]
    if len(methods) != 1:
        raise RuntimeError("BM25._process_document_batch not found")
    mod = ast.Module(body=methods, type_ignores=[])
    ast.fix_missing_locations(mod)
    ns = {"defaultdict": defaultdict, "List": List, "Dict": Dict, "Tuple": Tuple, "Optional": Optional}
    _exec(mod, ns, "<reference bm25_indexer.py>")
    return ns["_process_document_batch"]


def bm25_build_fixture(build_fn):
    """Three small crawls through the reference's own batch builder.  Tokeniser stand-in: str.split on the text the
    reference hands to it (lower-cased, city spellings unified, bm25_indexer.py:216-219) -- tokenisation itself (spaCy)
    stays unpinned.  Stored: the documents and the three tables the method returns."""
    def parallel(*_a, **_k):
        raise AssertionError("the fixture must stay below the 50-document switch to the multiprocessing branch")
    self = types.SimpleNamespace(_tokenize=lambda text: text.split(), _process_document_batch_parallel=parallel)
    rng = np.random.default_rng(77)
    words = ["schloss", "neckar", "tübingen", "castle", "market", "bridge", "museum", "garden", "tower", "boat", "river",
             "uni", "old", "town", "hall", "food", "beer", "walk", "hill", "view"]
    cases = []
    # (1) hand-made: a NULL title, a token-less document (only white space), both ASCII spellings of the city (also inside a
    #     longer word), upper case, a repeated term, ids not in insertion order
    docs1 = [(42, "Hohentubingen Castle", "The castle of Tuebingen above the Neckar castle"),
             (7, None, "TUBINGEN market MARKET Market"),
             (19, "   ", "  \t \n "),
             (3, "Boat", None),
             (100, "tübingen tuebingen tubingen", "x")]
    cases.append(("hand_made", docs1))
    # (2) random: 30 documents of 0..40 words from a 20-word vocabulary, ids with gaps, every fifth title NULL, two empty ones
    docs2 = []
    for i, doc_id in enumerate(sorted(rng.choice(5000, size=30, replace=False).tolist())):
        n = 0 if i in (4, 17) else int(rng.integers(1, 41))
        text = " ".join(rng.choice(words, size=n).tolist())
        title = None if i % 5 == 0 else " ".join(rng.choice(words, size=int(rng.integers(0, 4))).tolist())
        docs2.append((int(doc_id), title, text))
    cases.append(("random_30", docs2))
    # (3) 49 documents (the largest batch of the sequential branch), longer texts, ids descending
    docs3 = []
    for doc_id in sorted(rng.choice(100000, size=49, replace=False).tolist(), reverse=True):
        text = " ".join(rng.choice(words, size=int(rng.integers(20, 200))).tolist())
        docs3.append((int(doc_id), "Tuebingen " + str(doc_id % 7), text))
    cases.append(("batch_49_descending_ids", docs3))
    out = []
    for name, docs in cases:
        doc_stats, term_freq, term_updates = build_fn(self, docs)
        out.append({"name": name, "documents": [[d, t, x] for d, t, x in docs],
                    "doc_stats": [[int(d), int(l)] for d, l in doc_stats],
                    "term_freq": [[int(d), t, int(f)] for d, t, f in term_freq],
                    "term_updates": {t: [int(u["new_docs"]), int(u["freq_increase"])] for t, u in term_updates.items()}})
    dump_json("bm25_build.json", out)


def f32(x):
    return float(np.float32(x))


class _Result:
    def __init__(self, rows):
        self.rows = rows

    def fetchall(self):
        return self.rows

    def fetchone(self):
        return self.rows[0] if self.rows else None


class StubBM25Conn:
    """Answers the four SELECTs that BM25.search issues (bm25_indexer.py:374,413,436,494) from
    in-memory tables, applying the DuckDB behaviours listed in the module docstring."""

    def __init__(self, postings, doc_len, idf, avgdl, total_docs, urls_db):
        self.postings = postings      # term -> list[(doc_id, tf)]
        self.doc_len = doc_len        # doc_id -> int
        self.idf = idf                # term -> float (already float32-rounded) or None (NULL)
        self.avgdl, self.total_docs = avgdl, total_docs
        self.urls_db = urls_db        # doc_id -> (title|None, text)

    def execute(self, sql, params=()):
        s = " ".join(sql.split())
        if "FROM bm25_corpus_stats" in s:
            return _Result([("avg_doc_length", f32(self.avgdl)), ("total_docs", f32(self.total_docs))])
        if "FROM bm25_term_stats" in s:
            rows = []
            for t in dict.fromkeys(params):
                if t in self.postings:
                    p = self.postings[t]
                    rows.append((t, len(p), sum(tf for _, tf in p), self.idf[t]))
            return _Result(rows)
        if "FROM bm25_term_freq" in s:
            rows = set()
            for t in params:
                for d, tf in self.postings.get(t, []):
                    if d in self.doc_len:                      # inner JOIN bm25_doc_stats
                        rows.add((d, t, tf, self.doc_len[d]))
            return _Result(sorted(rows, key=lambda r: (r[0], r[1])))
        if "FROM urlsDB" in s:
            return _Result([(d, *self.urls_db[d]) for d in params if d in self.urls_db])
        raise AssertionError("unexpected SQL: " + s)


def run_bm25(search_fn, stats_fn, conn, terms, top_k, min_score, k1=1.2, b=0.75):
    self = types.SimpleNamespace(k1=k1, b=b, conn=conn)
    self._tokenize = lambda q: list(q)            # the query is handed over pre-tokenised
    self._get_corpus_stats = lambda: stats_fn(self)
    return search_fn(self, terms, top_k=top_k, min_score=min_score)


def idf_log10(N, df):
    return f32(math.log10((N - df + 0.5) / (df + 0.5)))


# --------------------------------------------------------------------------------------------
# fixture builders
# --------------------------------------------------------------------------------------------
def dump_json(name, obj):
    path = os.path.join(OUT, name)
    with open(path, "w", encoding="utf-8") as f:
        json.dump(obj, f, ensure_ascii=False, indent=None, separators=(",", ":"))
    print("wrote", name, os.path.getsize(path), "bytes")


def bm25_kat(search_fn, stats_fn):
    cases = []

    def add(name, N, postings, doc_len, urls_db, queries, idf_override=None, k1=1.2, b=0.75):
        avgdl = float(np.mean(list(doc_len.values())))
        idf = {t: idf_log10(N, len(p)) for t, p in postings.items()}
        if idf_override:
            idf.update(idf_override)
        conn = StubBM25Conn(postings, doc_len, idf, avgdl, N, urls_db)
        outs = []
        for (terms, top_k, min_score) in queries:
            res = run_bm25(search_fn, stats_fn, conn, terms, top_k, min_score, k1, b)
            outs.append({"terms": terms, "top_k": top_k, "min_score": min_score, "expected": res})
        cases.append({
            "name": name, "total_docs": N, "avgdl_f32": f32(avgdl), "k1": k1, "b": b,
            "postings": {t: [[d, tf] for d, tf in p] for t, p in postings.items()},
            "doc_len": {str(d): l for d, l in doc_len.items()},
            "idf_f32": idf,
            "urls_db": {str(d): [v[0], v[1]] for d, v in urls_db.items()},
            "queries": outs,
        })

    # 1. negative idf drowns everything
    post = {"castle": [(1, 2), (3, 1)], "tübingen": [(1, 1), (2, 3), (3, 1), (4, 1), (5, 2)]}
    dl = {1: 10, 2: 20, 3: 5, 4: 7, 5: 9, 6: 3}
    urls = {d: (f"T{d}", f"text of {d}") for d in dl}
    add("negative_idf_all_filtered", 6, post, dl, urls,
        [(["castle", "tübingen"], 3, 0.0), (["castle", "tübingen"], 3, -10.0), (["castle"], 3, 0.0),
         (["tübingen"], 10, -100.0)])
    # 2. ties, docs missing from urlsDB, NULL title, qtf, unknown term, truncation
    post = {"castle": [(3, 2), (7, 2), (11, 1)], "garden": [(3, 1), (7, 1), (20, 4)],
            "bridge": [(5, 1), (7, 3), (9, 1), (11, 1), (13, 1), (15, 1)]}
    dl = {d: 10 for d in (3, 5, 7, 9, 11, 13, 15, 20)}
    urls = {d: (f"Title {d}", "x" * 250 if d == 3 else f"body {d}") for d in dl if d != 20}
    urls[7] = (None, "seven")
    add("ties_missing_null", 1000, post, dl, urls,
        [(["castle", "garden"], 10, 0.0), (["castle", "castle", "garden"], 10, 0.0),
         (["unknown", "castle"], 10, 0.0), (["unknown"], 10, 0.0), ([], 10, 0.0),
         (["bridge"], 3, 0.0), (["bridge", "castle", "garden"], 2, 0.0),
         (["garden", "castle"], 10, 0.0), (["bridge"], 10, 0.9)])
    # 3. NULL idf -> 0.0 ; long docs vs short docs ; non-default k1/b
    post = {"a": [(1, 1), (2, 5), (3, 2)], "b": [(2, 1), (3, 1), (4, 9)], "nullidf": [(1, 3), (4, 1)]}
    dl = {1: 3, 2: 120, 3: 40, 4: 800}
    urls = {d: (f"t{d}", f"b{d}") for d in dl}
    add("null_idf_lengths", 50, post, dl, urls,
        [(["a", "b"], 10, 0.0), (["nullidf"], 10, 0.0), (["nullidf", "a"], 10, 0.0), (["b", "a", "b"], 10, 0.0)],
        idf_override={"nullidf": None})
    add("k1_b_variants", 50, post, dl, urls, [(["a", "b"], 10, 0.0), (["b"], 2, 0.0)], k1=2.0, b=0.3)
    dump_json("bm25_kat.json", cases)


def synth_corpus(seed, N, V, mean_len, city_frac=0.85):
    """Small Zipf corpus with one near-ubiquitous term (id 0, the stand-in for 'tübingen')."""
    rng = np.random.default_rng(seed)
    doc_ids = np.cumsum(rng.integers(1, 5, size=N)).astype(np.int64) + 100
    lens = np.clip(rng.lognormal(np.log(mean_len), 0.7, size=N), 4, 400).astype(np.int64)
    p = 1.0 / np.arange(1, V) ** 1.07
    p /= p.sum()
    rows = []
    for di in range(N):
        toks = rng.choice(np.arange(1, V), size=lens[di], p=p)
        if rng.random() < city_frac:
            toks[: max(1, int(rng.integers(1, 4)))] = 0
        t, c = np.unique(toks, return_counts=True)
        rows.append((t, c))
    post_d, post_t, post_tf = [], [], []
    for di, (t, c) in enumerate(rows):
        post_d.append(np.full(len(t), di)); post_t.append(t); post_tf.append(c)
    post_d, post_t, post_tf = map(np.concatenate, (post_d, post_t, post_tf))
    order = np.lexsort((post_d, post_t))
    post_d, post_t, post_tf = post_d[order], post_t[order], post_tf[order]
    df = np.bincount(post_t, minlength=V)
    term_off = np.zeros(V + 1, np.int64); term_off[1:] = np.cumsum(df)
    idf = np.array([idf_log10(N, int(d)) if d > 0 else 0.0 for d in df], np.float32)
    avgdl = np.float32(lens.mean())
    return dict(doc_ids=doc_ids, doc_len=lens.astype(np.int32), term_off=term_off,
                post_doc=post_d.astype(np.int32), post_tf=post_tf.astype(np.int32), idf=idf,
                avgdl=avgdl, total_docs=np.int64(N)), rng


def bm25_random(search_fn, stats_fn, name, seed, N, V, mean_len, nq):
    c, rng = synth_corpus(seed, N, V, mean_len)
    postings = {}
    for t in range(V):
        lo, hi = c["term_off"][t], c["term_off"][t + 1]
        if hi > lo:
            postings[f"t{t}"] = [(int(c["doc_ids"][d]), int(tf)) for d, tf in zip(c["post_doc"][lo:hi], c["post_tf"][lo:hi])]
    doc_len = {int(i): int(l) for i, l in zip(c["doc_ids"], c["doc_len"])}
    idf = {f"t{t}": float(c["idf"][t]) for t in range(V)}
    missing = set(int(x) for x in rng.choice(c["doc_ids"], size=max(1, N // 100), replace=False))
    urls = {d: (f"title {d}", f"text {d}") for d in doc_len if d not in missing}
    conn = StubBM25Conn(postings, doc_len, idf, float(c["avgdl"]), N, urls)
    df = np.diff(c["term_off"])
    cand = np.nonzero(df > 0)[0]
    w = df[cand].astype(np.float64) ** 0.5
    w /= w.sum()
    queries = []
    for qi in range(nq):
        nt = int(rng.integers(1, 5))
        terms = [int(x) for x in rng.choice(cand, size=nt, replace=False, p=w)]
        if qi % 3 != 2:
            terms.append(0)                       # the forced city term, like preprocess_query does
        if qi % 5 == 4:
            terms.insert(1, terms[0])             # repeated term => qtf 2
        if qi % 7 == 6:
            terms.insert(0, V + 17)               # unknown term
        rng.shuffle(terms)
        for (top_k, min_score) in ((10, 0.0), (100, 0.0), (1000, 0.0)) + (((100, -4.0),) if qi % 4 == 0 else ()) + (((100, 1.5),) if qi % 4 == 1 else ()):
            res = run_bm25(search_fn, stats_fn, conn, [f"t{t}" for t in terms], top_k, min_score)
            queries.append({"terms": terms, "top_k": top_k, "min_score": min_score,
                            "doc_id": [r["doc_id"] for r in res], "score": [r["score"] for r in res]})
    np.savez_compressed(os.path.join(OUT, f"{name}.npz"), **c)
    dump_json(f"{name}.json", {"missing_from_urlsdb": sorted(missing), "queries": queries})


def unit_rows(rng, n, d=768):
    x = rng.standard_normal((n, d)).astype(np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True).astype(np.float32)
    return x.astype(np.float32)


def cosine_fixture(ns):
    import pandas as pd
    rng = np.random.default_rng(7)
    E = unit_rows(rng, 1000)
    E[5] *= 3.0                       # rows that are not unit norm: sklearn re-normalises them
    E[6] *= 0.01
    E[7] = 0.0                        # zero row: sklearn divides by 1 instead of 0
    q = (E[123] + 0.5 * unit_rows(rng, 1)[0]).astype(np.float32)
    q = (q / np.linalg.norm(q) * 9.5).astype(np.float32)        # encoder output is not normalised
    df = pd.DataFrame({"embedding": list(E)})
    sims = ns["get_new_similarity"](df, q, 32)
    one = ns["calculate_similarity"](list(map(float, q)), list(map(float, E[123])))
    np.savez_compressed(os.path.join(OUT, "cosine.npz"), q=q, E=E,
                        expected=np.asarray(sims, dtype=np.float32),
                        expected_dtype=str(np.asarray(sims).dtype), single_123=np.float64(one))
    print("wrote cosine.npz", "dtype", np.asarray(sims).dtype)


def cosine_unit_fixture(ns):
    """All rows unit-norm (what indexer/indexer.py:165 stores), documents of 2..12 chunks, 40 queries: the shape on
    which the engine's DEFAULT dense kernel (f16x2-split products) runs, so that kernel is compared with values the
    reference's own get_new_similarity (reranker_api.py:273-287) produced.  expected[i][c] = cosine(query i, chunk c)."""
    import pandas as pd
    rng = np.random.default_rng(23)
    n_chunks_of = rng.integers(2, 13, size=140)
    doc_off = np.zeros(len(n_chunks_of) + 1, np.int32)
    doc_off[1:] = np.cumsum(n_chunks_of)
    E = unit_rows(rng, int(doc_off[-1]))
    Q = 40
    qs = np.empty((Q, 768), np.float32)
    for i in range(Q):
        base = E[int(rng.integers(0, len(E)))] if i % 2 == 0 else unit_rows(rng, 1)[0]
        v = (base + 0.5 * unit_rows(rng, 1)[0]).astype(np.float32)
        qs[i] = (v / np.linalg.norm(v) * rng.uniform(0.5, 15.0)).astype(np.float32)   # encoder output is not normalised
    df = pd.DataFrame({"embedding": list(E)})
    exp = np.stack([np.asarray(ns["get_new_similarity"](df, qs[i], 32), dtype=np.float32) for i in range(Q)])
    np.savez_compressed(os.path.join(OUT, "cosine_unit.npz"), q=qs, E=E, doc_off=doc_off, expected=exp)
    print("wrote cosine_unit.npz", exp.shape, exp.dtype)


def rerank_fixtures(ns):
    rng = np.random.default_rng(11)
    cases = []
    for ci, (ndocs, maxch) in enumerate(((4, 4), (40, 14), (120, 12))):
        doc_ids = np.sort(rng.choice(np.arange(10, 10 + ndocs * 6), size=ndocs, replace=False))
        doms = [f"site{j}.example.de" for j in range(max(3, ndocs // 4))]
        urls, chunks, emb = [], [], {}
        cid = 1000
        base = unit_rows(rng, 1)[0]
        for d in doc_ids:
            dom = doms[int(rng.integers(0, len(doms)))]
            url = f"https://{dom}/page/{int(d)}"
            if ci > 0 and d % 11 == 0:
                url += "?q=vf"
            title = None if (ci > 0 and d % 13 == 0) else f"Title {int(d)}"
            urls.append((int(d), url, title, f"full text of {int(d)} " * 3))
            nch = int(rng.integers(1, maxch + 1))
            for _ in range(nch):
                v = unit_rows(rng, 1)[0] + float(rng.uniform(0, 1.2)) * base
                emb[cid] = (v / np.linalg.norm(v)).astype(np.float32)
                chunks.append((cid, int(d)))
                cid += 1
        if ci > 0:
            # two ids that share a URL modulo the query string -> only MIN(id) survives
            a = urls[3]
            dup_id = int(doc_ids.max()) + 5
            urls.append((dup_id, a[1].split("?")[0] + "?session=1", "dup", "dup text"))
            emb[cid] = unit_rows(rng, 1)[0]; chunks.append((cid, dup_id)); cid += 1
        db = StubRerankDB(urls, chunks, emb)
        req_ids = [u[0] for u in urls]
        if ci > 0:
            req_ids.append(999999)                 # requested but not in urlsDB
        order = rng.permutation(len(req_ids))
        req_ids = [req_ids[i] for i in order]
        sims = [float(x) for x in np.sort(rng.uniform(0.2, 14.0, size=len(req_ids)))[::-1]]
        q = (base * 7.0 + unit_rows(rng, 1)[0] * 2.0).astype(np.float32)
        stages = run_rerank_stages(ns, db, q, req_ids, sims)
        out = {}
        for div in (True, False):
            resp = run_rerank(ns, db, q, req_ids, sims, diversification=div)
            out["div" if div else "nodiv"] = {
                "document_scores": [{"doc_id": d.doc_id, "title": d.title, "url": d.url,
                                     "similarity_score": d.similarity_score,
                                     "original_similarity": d.original_similarity,
                                     "window_index": d.most_relevant_window.window_index}
                                    for d in resp.document_scores],
                "top_windows": [{"doc_id": w.doc_id, "window_index": w.window_index,
                                 "similarity_score": w.similarity_score} for w in resp.top_windows],
                "total_documents": resp.total_documents, "total_windows": resp.total_windows}
        cids = sorted(emb)
        np.savez_compressed(os.path.join(OUT, f"rerank_{ci}.npz"), q=q,
                            chunk_id=np.array(cids, np.int64),
                            chunk_doc=np.array([dict(chunks)[c] for c in cids], np.int64),
                            emb=np.stack([emb[c] for c in cids]).astype(np.float32))
        cases.append({"case": ci, "urls": [list(u) for u in urls], "doc_ids": req_ids,
                      "similarities": sims, "stages": stages, "response": out})
    # empty result -> HTTP 401 (reranker_api.py:348-349)
    db = StubRerankDB([], [], {})
    try:
        run_rerank(ns, db, np.ones(768, np.float32), [1, 2], [1.0, 0.5])
        err = None
    except Exception as e:  # noqa
        err = {"type": type(e).__name__, "status_code": getattr(e, "status_code", None)}
    dump_json("rerank_chain.json", {"cases": cases, "empty_error": err})


def diversification_fixture(ns):
    DS, WS = ns["DocumentScore"], ns["WindowScore"]

    def mk(i, url, s):
        w = WS(text="t", similarity_score=s, doc_id=str(i), title="t", window_index=0)
        return DS(doc_id=str(i), title="t", url=url, similarity_score=s, original_similarity=0.0, most_relevant_window=w)

    rng = np.random.default_rng(5)
    cases = []
    inputs = [[(1, "http://a.de/x", .95), (2, "http://a.de/y", .90), (3, "http://b.de/", .85),
               (4, "http://c.de/1", .50), (5, "http://c.de/2", .40), (6, "http://a.de/z", .30)]]
    for n, nd in ((30, 4), (150, 20), (400, 300), (12, 12)):
        sc = np.sort(rng.uniform(0, 1, size=n))[::-1]
        inputs.append([(i + 1, f"https://D{int(rng.integers(0, nd))}.org/p{i}", float(s)) for i, s in enumerate(sc)])
    for inp in inputs:
        for top_k in (5, 100):
            docs = [mk(*r) for r in inp]
            res = ns["hybrid_diversification"](docs, top_k=top_k)
            cases.append({"input": [list(r) for r in inp], "top_k": top_k,
                          "expected": [[int(d.doc_id), d.similarity_score] for d in res]})
    capped = []
    for inp in inputs[:3]:
        docs = [mk(*r) for r in inp]
        kept, dropped = ns["apply_domain_cap"](docs, max_per_domain=2)
        capped.append({"input": [list(r) for r in inp], "kept": [int(d.doc_id) for d in kept],
                       "dropped": [int(d.doc_id) for d in dropped]})
    doms = ["https://www.Uni-Tuebingen.de/a?b=1", "http://tuebingen.de", "not a url", "", "ftp://x.y.z:21/q"]
    dump_json("diversification.json", {"cases": cases, "domain_cap": capped,
                                       "extract_domain": [[u, ns["extract_domain"](u)] for u in doms]})


def windows_fixture(ns):
    out = []
    for n in (0, 10, 512, 513, 900, 962, 1000, 1412, 1413, 2000):
        for (w, s) in ((512, 450), (8, 3)):
            wins = ns["create_sliding_windows"](list(range(n)), w, s)
            out.append({"n": n, "window": w, "step": s, "starts": [x[0] if x else -1 for x in wins],
                        "lens": [len(x) for x in wins]})
    dump_json("windows.json", out)


def search_api_fixture():
    import re
    from urllib.parse import urlparse
    tree = ast.parse(_read("search_api.py"))
    ns = {"re": re, "urlparse": urlparse, "logging": logging}
    _exec(_pick(tree, ["preprocess_query", "extract_domain_topic"]), ns, "<reference search_api.py>")
    qs = []
    with open(os.path.join(REF, "queries.txt"), encoding="utf-8") as f:
        for line in f:
            parts = line.strip().split("\t")
            if len(parts) >= 2:
                qs.append(parts[1].strip())
    qs += ["  Tuebingen Castle ", "TUBINGEN tuebingen", "Neckar", "", "hohentübingen", "Tübingen"]
    urls = ["https://www.uni-tuebingen.de/en/", "https://tuebingen.de", "http://a.b.c.example.co.uk/x", "#", "",
            "https://localhost:5000/", "https://www.my_site!.de/"]
    dump_json("search_api.json", {"preprocess_query": [[q, ns["preprocess_query"](q)] for q in qs],
                                  "extract_domain_topic": [[u, ns["extract_domain_topic"](u)] for u in urls]})


def main():
    if not os.path.isdir(REF):
        sys.exit("reference tree not present; goldens can only be regenerated in the build container")
    logging.disable(logging.CRITICAL)
    search_fn, stats_fn = load_bm25_methods()
    bm25_kat(search_fn, stats_fn)
    bm25_random(search_fn, stats_fn, "bm25_random_a", seed=101, N=600, V=400, mean_len=30, nq=14)
    bm25_random(search_fn, stats_fn, "bm25_random_b", seed=202, N=2500, V=1500, mean_len=45, nq=20)
    bm25_build_fixture(load_bm25_build_method())
    ns = load_reranker_namespace()
    cosine_fixture(ns)
    cosine_unit_fixture(ns)
    rerank_fixtures(ns)
    diversification_fixture(ns)
    windows_fixture(ns)
    search_api_fixture()


if __name__ == "__main__":
    main()
