"""The ``compile_context`` slice of the reference's ``BatteryNode`` (lb/node.py:718-734).

Everything else a BatteryNode does (chain, CAS, P2P, keys, WAL) is out of scope
(SURVEY.md §2); a maintainer wires the GPU path in by swapping ``lb.context_graph``
(INTEGRATION.md).  This class exists so the node-level contract -- unknown group
raises ``NodeError``, otherwise exactly ``graph.compile(...)`` -- is testable here.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Any, Dict, List, Optional, Tuple

from .context_graph import ContextGraph


class NodeError(Exception):  # lb/node.py:33
    pass


@dataclass
class Group:
    """Holds what compile_context touches of lb/group.py:16 -- the graph."""
    group_id: str
    graph: ContextGraph = field(default_factory=ContextGraph)


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
                        since_ms: Optional[int] = None) -> Tuple[str, List[str]]:
        """lb/node.py:718-734: a pure delegate; takes no lock, like the reference."""
        g = self._require_group(group_id)
        slice_text, chosen = g.graph.compile(query, top_k=top_k, since_ms=since_ms)
        return slice_text, chosen

    def get_recent_claims(self, group_id: str, since_ms: int, *, limit: int = 100) -> List[Dict[str, Any]]:
        """lb/node.py:736-762: claims created at or after ``since_ms``, newest first, at most ``limit``.
        The filter + sort + cut run on the GPU (``ContextGraph.recent``); only the chosen claims are
        turned into dicts here."""
        g = self._require_group(group_id)
        claims = g.graph.claims
        out = []
        for h in g.graph.recent(since_ms, limit):
            c = claims[h]
            out.append({
                "claim_hash": h,
                "text": c.text,
                "tags": list(c.tags),
                "created_ms": c.created_ms,
                "parent_hash": c.parent_hash,
            })
        return out
