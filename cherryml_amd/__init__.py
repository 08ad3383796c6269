"""cherryml_amd: MI355X-native composite-likelihood core with CherryML's API
surface for that path (see DESIGN.md / INTEGRATION.md)."""
from .bank import CherryBank  # noqa: F401

__all__ = ["CherryBank"]
