"""HTTP facades with the reference's routes and JSON shapes (plumbing, no arithmetic):

    POST /rerank             reranker/reranker_api.py:336-417   (RerankRequest -> RerankResponse, 401 / 500)
    POST /api/search         search_api.py:69-152               ({llm_response, documents:[...]})
    POST /api/batch_search   search_api.py:204-328              (queries.txt -> qnum<TAB>rank<TAB>url<TAB>score lines)
    POST /api/batch_search_file  search_api.py:331-367          (the same, written to batch_search_results.txt)
    GET  /api/health         search_api.py:369-375
    GET  /                   search_api.py:377-380              (the UI page: templates/index.html of the deployment when
                                                                 `ui_dir` is given -- the reference's D3 front end is not
                                                                 part of this build -- else a one-line placeholder)

The reference runs two processes (Flask :5000 + FastAPI :8000) that talk JSON over HTTP; here both sets of
routes sit on one FastAPI app over one Retriever.  The LLM summariser (search_assistant/, a cloud call) is
out of scope: `llm` is an optional callable(query, windows) -> str, otherwise llm_response is "".
Requests may carry `query_embedding` (768 floats) and `terms` (pre-tokenised query) for deployments that
keep the encoder / spaCy in another process.
"""
import uuid
from typing import List, Optional

from .reranker import RerankNotFound
from .text import preprocess_query, read_queries_file

LLM_MAX_WINDOWS = 10          # config.py:22


def create_app(retriever, llm=None, queries_file="queries.txt", results_file="batch_search_results.txt", ui_dir=None):
    import os

    from fastapi import FastAPI
    from fastapi.responses import HTMLResponse, JSONResponse
    from pydantic import BaseModel

    class RerankRequest(BaseModel):
        doc_ids: List[str]
        similarities: Optional[List[float]] = None
        query: str
        query_embedding: Optional[List[float]] = None

    class SearchRequest(BaseModel):
        query: str = ""
        top_k: int = 1000
        query_id: Optional[str] = None
        query_embedding: Optional[List[float]] = None
        terms: Optional[List[str]] = None

    app = FastAPI(title="Document Reranker API", version="1.0.0")

    @app.post("/rerank")
    def rerank(req: RerankRequest):
        try:
            return retriever.reranker.rerank(req.doc_ids, req.similarities, query=req.query,
                                             query_embedding=req.query_embedding)
        except RerankNotFound as e:
            return JSONResponse(status_code=401, content={"detail": str(e)})
        except Exception as e:
            return JSONResponse(status_code=500, content={"detail": f"Internal server error: {e}"})

    @app.post("/api/search")
    def search(req: SearchRequest):
        try:
            query = preprocess_query(req.query.strip())
            if not query:
                return JSONResponse(status_code=400, content={"error": "Query is required"})
            qid = req.query_id or uuid.uuid4().hex
            docs = retriever.search(req.query, top_k=req.top_k, query_embedding=req.query_embedding,
                                    terms=req.terms, query_id=qid)
            llm_response = ""
            if llm is not None and docs:
                llm_response = llm(query, [d["snippet"] for d in docs[:LLM_MAX_WINDOWS]])
            return {"llm_response": llm_response, "documents": docs}
        except Exception:
            return JSONResponse(status_code=500, content={"error": "Internal server error"})

    def _batch():
        """-> (status, body): the body of /api/batch_search (search_api.py:204-328)."""
        try:
            try:
                queries = read_queries_file(queries_file)
            except FileNotFoundError:
                return 404, {"error": "queries.txt file not found"}
            if not queries:
                return 400, {"error": "No valid queries found in queries.txt"}
            results = retriever.batch_search(queries)
            return 200, {"total_queries": len(queries), "total_results": len(results), "results": list(results), "_lines": results,
                         "queries_processed": [{"query_num": n, "query_text": t} for n, t in queries]}
        except Exception as e:
            return 500, {"error": f"Internal server error: {e}"}

    @app.post("/api/batch_search")
    def batch_search():
        status, body = _batch()
        body.pop("_lines", None)
        return body if status == 200 else JSONResponse(status_code=status, content=body)

    @app.post("/api/batch_search_file")
    def batch_search_file():
        # search_api.py:331-367: run the batch search, pass its error through unchanged, else write one formatted line per
        # result and report where they went
        try:
            status, body = _batch()
            if status != 200:
                return JSONResponse(status_code=status, content=body)
            lines = body.get("_lines")
            if hasattr(lines, "write"):                       # Retriever.batch_search: all lines formatted natively in one call
                lines.write(results_file)
            else:
                with open(results_file, "w", encoding="utf-8") as f:
                    for r in body["results"]:
                        f.write(r["formatted_line"] + "\n")
            return {"message": f"Results saved to {results_file}", "total_queries": body["total_queries"],
                    "total_results": body["total_results"], "output_file": str(results_file),
                    "format": "query_num<tab>rank<tab>url<tab>score per line"}
        except Exception as e:
            return JSONResponse(status_code=500, content={"error": f"Internal server error: {e}"})

    @app.get("/", response_class=HTMLResponse)
    def index():
        page = os.path.join(ui_dir, "templates", "index.html") if ui_dir else None
        if page and os.path.exists(page):
            with open(page, encoding="utf-8") as f:
                return f.read()
        return "<html><body><p>msretr search API: POST /api/search, /api/batch_search, /api/batch_search_file, /rerank</p></body></html>"

    @app.get("/api/health")
    def health():
        return {"status": "healthy", "search_engine_ready": retriever is not None}

    return app
