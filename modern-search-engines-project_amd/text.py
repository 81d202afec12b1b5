"""Host-side text helpers of the query path (strings never reach the GPU).

Mirrors, by behaviour, /root/reference/search_api.py:155-166 (preprocess_query), :168-201
(extract_domain_topic), reranker/reranker_api.py:170-176 (extract_domain), :239-260
(create_sliding_windows) and the batch line format of search_api.py:290.
"""
import re
from urllib.parse import urlparse

CITY = "tübingen"


def preprocess_query(query: str) -> str:
    """Lower-case, map the ASCII spellings of the city to 'tübingen', append the city when the query does
    not mention it (search_api.py:155-166)."""
    q = query.strip().lower()
    if "tuebingen" in q or "tubingen" in q or CITY in q:
        q = q.replace("tuebingen", CITY).replace("tubingen", CITY)
    else:
        q = f"{q} {CITY}"
    return q.replace("tuebingen", CITY).replace("tubingen", CITY).strip().lower()


def extract_domain(url) -> str:
    try:
        return urlparse(url).netloc.lower()
    except Exception:
        return "defaultdomain"


def extract_domain_topic(url) -> str:
    if not url or url == "#":
        return "unknown"
    try:
        domain = re.sub(r"^www\.", "", urlparse(url).netloc.lower())
        parts = domain.split(".")
        main = (parts[0] if len(parts) == 2 else parts[-2]) if len(parts) >= 2 else domain
        main = re.sub(r"[^a-zA-Z0-9-]", "", main)
        return main if main else "unknown"
    except Exception:
        return "unknown"


def create_sliding_windows(tokens, window_size=512, step_size=450):
    """Token windows used to cut documents into chunks (config.py:10-11; embedder.py:65-87)."""
    if len(tokens) <= window_size:
        return [tokens]
    windows = [tokens[i:i + window_size] for i in range(0, len(tokens) - window_size + 1, step_size)]
    last = len(tokens) - window_size
    if last >= 0 and last % step_size != 0:
        windows.append(tokens[last:last + window_size])
    return windows


_WORD = re.compile(r"[^\W\d_]+", re.UNICODE)


def simple_tokenize(text: str):
    """Stand-in for BM25._tokenize (bm25_indexer.py:149-155).  The reference lemmatises with spaCy
    `en_core_web_sm` and drops stop words; spaCy and its model are not available offline, so this only
    lower-cases and keeps alphabetic tokens.  Pass `tokenizer=` to BM25 / Retriever to plug in the real
    one; with identical term lists the engine's results are identical to the reference's."""
    return [m.group(0).lower() for m in _WORD.finditer(text)]


def format_result_line(query_num, rank, url, score) -> str:
    """search_api.py:290"""
    return f"{query_num}\t{rank}\t{url}\t{score:.3f}"


def read_queries_file(path):
    """queries.txt: 'query_num<TAB>query_text' per line (search_api.py:214-235)."""
    out = []
    with open(path, "r", encoding="utf-8") as f:
        for line in f:
            line = line.strip()
            if not line:
                continue
            parts = line.split("\t")
            if len(parts) >= 2:
                out.append((parts[0].strip(), parts[1].strip()))
    return out


class LineFormatter:
    """The batch lines of search_api.py:290 for whole batches: the URLs as one UTF-8 blob + offsets (built once), the
    formatting in native code (msr_format_lines, include/msretr.h: a HOST function, no GPU involved).  Byte for byte what
    "\n".join(format_result_line(...)) + "\n" gives."""

    def __init__(self, urls, n_docs=None):
        import numpy as np
        from . import _abi
        self._lib = _abi.load()
        n = len(urls) if urls is not None else int(n_docs or 0)
        enc = [u.encode("utf-8") if u else b"" for u in urls] if urls is not None else []
        self.blob = b"".join(enc)
        self.off = np.zeros(n + 1, np.int64)
        if enc:
            np.cumsum(np.fromiter((len(b) for b in enc), np.int64, n), out=self.off[1:])
        self.n_docs = n
        self.max_url = int(np.diff(self.off).max()) if n else 0
        self._buf = self._cbuf = None

    def format(self, query_nums, doc, score, n):
        """query_nums: list of str; doc int32 [Q, S], score float64 [Q, S], n int32 [Q] (host arrays) -> the lines as a
        bytes-like view of the formatter's own buffer (valid until the next call; bytes(...) for a copy)."""
        import ctypes as C

        import numpy as np
        doc = np.ascontiguousarray(doc, np.int32); score = np.ascontiguousarray(score, np.float64)
        n = np.ascontiguousarray(n, np.int32)
        Q = len(query_nums)
        assert doc.shape == score.shape and doc.ndim == 2 and doc.shape[0] == Q and n.shape == (Q,)
        qb = [str(x).encode("utf-8") for x in query_nums]
        qblob = b"".join(qb)
        qoff = np.zeros(Q + 1, np.int64)
        if Q:
            np.cumsum(np.fromiter((len(b) for b in qb), np.int64, Q), out=qoff[1:])
        ptr = lambda a: C.c_void_p(a.ctypes.data)
        args = (C.c_char_p(qblob), ptr(qoff), Q, ptr(doc), ptr(score), ptr(n), int(doc.shape[1]), C.c_char_p(self.blob),
                ptr(self.off), self.n_docs, max(1, self.max_url))
        need = -int(self._lib.msr_format_lines(*args, None, 0))
        if need <= 0:
            return memoryview(b"")
        if self._buf is None or len(self._buf) < need:         # one output buffer per formatter, grown when needed (a fresh
            self._buf = bytearray(need + need // 4)             # bytearray of several MB costs more than filling it)
            self._cbuf = (C.c_char * len(self._buf)).from_buffer(self._buf)
        got = int(self._lib.msr_format_lines(*args, self._cbuf, len(self._buf)))
        if got < 0:
            raise RuntimeError(f"msr_format_lines failed ({got})")
        return memoryview(self._buf)[:got]
