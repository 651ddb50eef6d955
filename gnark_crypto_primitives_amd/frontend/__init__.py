from .api import API, Variable, CompileError  # noqa: F401
from .compile import Public, Secret, CompiledCircuit, compile_circuit  # noqa: F401
