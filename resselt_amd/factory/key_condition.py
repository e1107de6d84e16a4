"""Checkpoint key predicates used for architecture sniffing.

Same contract as the reference ``resselt/factory/key_condition.py:6-32``: a condition is a tree whose
leaves are key names (satisfied when the key is present in the state dict) and whose inner nodes are
``has_all`` / ``has_any`` combinators.
"""

from __future__ import annotations

from typing import Literal, Mapping, Union

Kind = Literal['all', 'any']
Clause = Union[str, 'KeyCondition']


class KeyCondition:
    def __init__(self, kind: Kind, keys: tuple[Clause, ...]):
        if kind not in ('all', 'any'):
            raise ValueError(f"kind must be 'all' or 'any', got {kind!r}")
        self._kind: Kind = kind
        self._keys: tuple[Clause, ...] = tuple(keys)

    @staticmethod
    def has_all(*keys: Clause) -> 'KeyCondition':
        return KeyCondition('all', keys)

    @staticmethod
    def has_any(*keys: Clause) -> 'KeyCondition':
        return KeyCondition('any', keys)

    def _holds(self, clause: Clause, state_dict: Mapping[str, object]) -> bool:
        return clause(state_dict) if isinstance(clause, KeyCondition) else clause in state_dict

    def __call__(self, state_dict: Mapping[str, object]) -> bool:
        results = (self._holds(c, state_dict) for c in self._keys)
        return all(results) if self._kind == 'all' else any(results)

    def __repr__(self) -> str:
        return f'KeyCondition.has_{self._kind}({", ".join(map(repr, self._keys))})'
