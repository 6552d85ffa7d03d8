"""primate_amd — MI355X-native stochastic Lanczos quadrature behind primate's operator/plugin API."""

__version__ = "0.1.0"


def get_include() -> str:
	"""Directory of the C header of the native boundary (`slq.h`), for extension modules that link against
	`primate_amd/_libslq.so`; the counterpart of `primate.get_include()` (src/primate/__init__.py:17-41), whose
	header-only C++ concept is replaced by the C-ABI."""
	import os

	return os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include")
