"""Drop-in for humanrf.scene_representation.tensor_composition_native
(humanrf/scene_representation/native/tensor_composition.cu:120-225): same two functions, same argument order,
same outputs. The fused encoder (hrf_encode4d_*) never calls these; they exist so code written against the
reference's op keeps working."""
from ..ops import compose_backward as compose_tensors_backward  # noqa: F401
from ..ops import compose_forward as compose_tensors_forward  # noqa: F401
