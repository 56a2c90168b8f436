"""Error types (mirror of the reference's pvtrace/common/errors.py:1-13)."""


class AppError(Exception):
    """Base class for errors raised by the scene API."""


class TraceError(AppError):
    """The tracer reached an inconsistent state."""


class GeometryError(AppError):
    """A geometry query could not be answered."""
