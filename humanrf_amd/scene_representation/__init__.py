from .decomposition4d import Decomposition4D  # noqa: F401
from .humanrf import HumanRF  # noqa: F401
from .query_io import QueryInput, QueryOutput  # noqa: F401
