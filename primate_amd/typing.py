"""Keyword-argument helpers of `primate.typing` (src/primate/typing.py:5-15)."""

from __future__ import annotations

import inspect
from typing import Callable


def restrict_kwargs(fun: Callable, kwargs) -> dict:
	"""The entries of `kwargs` that `fun` accepts as parameters."""
	accepted = inspect.signature(fun).parameters
	return {k: v for k, v in kwargs.items() if k in accepted}


def setdiff_kwargs(f: Callable, kwargs) -> dict:
	"""The entries of `kwargs` that `f` does NOT accept as parameters."""
	accepted = inspect.signature(f).parameters
	return {k: v for k, v in kwargs.items() if k not in accepted}
