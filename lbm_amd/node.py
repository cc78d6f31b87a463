"""The ``compile_context`` slice of the reference's ``BatteryNode`` (lb/node.py:718-734).

Everything else a BatteryNode does (chain, CAS, P2P, keys, WAL) is out of scope
(SURVEY.md §2); a maintainer wires the GPU path in by swapping ``lb.context_graph``
(INTEGRATION.md).  This class exists so the node-level contract -- unknown group
raises ``NodeError``, otherwise exactly ``graph.compile(...)`` -- is testable here.
"""
from __future__ import annotations

import json
import os
from dataclasses import dataclass, field
from typing import Any, Dict, Iterable, List, Optional, Sequence, Tuple

from .context_graph import ContextGraph


class NodeError(Exception):  # lb/node.py:33
    pass


@dataclass
class Group:
    """Holds what compile_context touches of lb/group.py:16 -- the graph -- and the graph half of its
    save / load (lb/group.py:30-55: ``graph.json`` = ``ContextGraph.snapshot()``; chain.json is out of scope)."""
    group_id: str
    graph: ContextGraph = field(default_factory=ContextGraph)

    def save_graph(self, root: str) -> str:
        """lb/group.py:40-44 for the graph: ``<root>/graph.json``, written to a temporary file and renamed."""
        os.makedirs(root, exist_ok=True)
        path = os.path.join(root, "graph.json")
        tmp = path + ".tmp"
        with open(tmp, "w", encoding="utf-8") as f:
            json.dump(self.graph.snapshot(), f)
        os.replace(tmp, path)
        return path

    @staticmethod
    def load_graph(group_id: str, root: str) -> "Group":
        """lb/group.py:47-55 for the graph: ``from_snapshot`` of graph.json when it exists, else an empty graph.  The
        reference re-embeds every live claim here, one Python ``embed`` at a time; this mirror defers it to ONE K1
        launch at the first query (lbm_amd/context_graph.py:from_snapshot)."""
        path = os.path.join(root, "graph.json")
        if os.path.exists(path):
            with open(path, "r", encoding="utf-8") as f:
                return Group(group_id, ContextGraph.from_snapshot(json.load(f)))
        return Group(group_id)


def _claim_fields(obj: Dict[str, Any]) -> Dict[str, Any]:
    """Keyword arguments of ``add_claim`` from a stored claim object, with the reference's defaults and coercions
    (lb/node.py:499-506: missing text -> "", missing lists -> [], created_ms None / missing -> 0)."""
    created = obj.get("created_ms", 0)
    return {"text": str(obj.get("text", "")),
            "tags": list(obj.get("tags", [])),
            "evidence": list(obj.get("evidence", [])),
            "created_ms": int(created) if created else 0,
            "parent_hash": obj.get("parent_hash")}


def _replay_ops(txs: Iterable[Dict[str, Any]]):
    """(op, claim_hash, stored object | None) for every transaction that changes a graph, in stream order.
    lb/node.py:493-513: a claim needs a ``str`` hash that the CAS holds -- here: the transaction carries an ``object``
    key at all -- and ``cas.get_json(h) or {}`` turns every falsy object (None, {}, [], 0, "") into an EMPTY claim;
    what the object is worth is decided where the claim is added (``rebuild_group_graph``)."""
    for tx in txs:
        kind = tx.get("type")
        if kind == "retract":
            target = tx.get("claim_hash")
            if isinstance(target, str):
                yield "retract", target, None
        elif kind == "claim":
            claim_hash = tx.get("artifact_hash")
            if isinstance(claim_hash, str) and "object" in tx:
                yield "claim", claim_hash, (tx["object"] or {})


class RetrievalNode:
    def __init__(self) -> None:
        self.groups: Dict[str, Group] = {}

    def create_group(self, group_id: str) -> str:
        self.groups[group_id] = Group(group_id)
        return group_id

    def _require_group(self, group_id: str) -> Group:  # lb/node.py:1238-1242
        g = self.groups.get(group_id)
        if not g:
            raise NodeError(f"unknown group_id {group_id}")
        return g

    def compile_context(self, group_id: str, query: str, *, top_k: int = 8,
                        since_ms: Optional[int] = None, lazy: bool = False) -> Tuple[str, List[str]]:
        """lb/node.py:718-734: a pure delegate; takes no lock, like the reference.  ``lazy=True`` (extension): a
        ``LazyContext`` whose text is formatted on first use."""
        g = self._require_group(group_id)
        if lazy:
            return g.graph.compile(query, top_k=top_k, since_ms=since_ms, lazy=True)
        slice_text, chosen = g.graph.compile(query, top_k=top_k, since_ms=since_ms)
        return slice_text, chosen

    def rebuild_group_graph(self, group_id: str, txs: Iterable[Dict[str, Any]]) -> None:
        """lb/node.py:486-515 without the chain and the CAS (out of scope): ``txs`` is the group's transaction stream
        with the claim objects already resolved -- ``{"type": "claim", "artifact_hash": h, "object": {...}}`` (what
        ``cas.get_json(h)`` returns; no ``object`` key = a CAS miss, skipped; a falsy object = an empty claim; anything that
        raises while it is read -- a truthy non-dict, ``created_ms`` = Infinity, tags that are not iterable -- skips the claim) and
        ``{"type": "retract", "claim_hash": h}``.  A NEW graph replaces the group's, exactly as in the reference; every
        claim it keeps is vectorised in one K1 launch at the first query and the retracts never reach the device."""
        group = self._require_group(group_id)
        fresh = ContextGraph()
        for op, claim_hash, obj in _replay_ops(txs):
            if op == "retract":
                fresh.retract(claim_hash)
                continue
            try:  # lb/node.py:497-509: ANY exception while rebuilding one claim skips that claim and nothing else
                fresh.add_claim(claim_hash, **_claim_fields(obj))
            except Exception:
                continue
        group.graph = fresh

    def compile_context_many(self, group_id: str, queries: Sequence[str], *, top_k: int = 8,
                             since_ms: Optional[int] = None, lazy: bool = False) -> List[Tuple[str, List[str]]]:
        """Batched ``compile_context`` (the reference has no batched entry -- its callers, e.g. the
        MCP tool at lb/mcp.py:271-279, loop): same group check, then every query is ranked in ONE pass over the group's
        claim matrix on the matrix cores (K3).  Element i equals ``compile_context(group_id, queries[i], ...)``.
        ``lazy=True`` returns ``LazyContext`` objects: claim hashes at once, the context text formatted when it is read
        (lb/context_graph.py:122-133 leaves the query's critical path)."""
        g = self._require_group(group_id)
        return g.graph.compile_many(queries, top_k=top_k, since_ms=since_ms, lazy=lazy)

    def get_recent_claims(self, group_id: str, since_ms: int, *, limit: int = 100) -> List[Dict[str, Any]]:
        """lb/node.py:736-762: claims created at or after ``since_ms``, newest first, at most ``limit``.
        The filter + sort + cut run on the GPU (``ContextGraph.recent``); only the chosen claims are turned into
        dicts (``ContextGraph.recent_claims``; a claim-sharded graph gathers them from their owners)."""
        g = self._require_group(group_id)
        return g.graph.recent_claims(since_ms, limit)
