"""primate_amd — MI355X-native stochastic Lanczos quadrature behind primate's operator/plugin API."""

__version__ = "0.1.0"
