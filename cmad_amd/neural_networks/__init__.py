from .input_convex_neural_network import AffineScaler, InputConvexNeuralNetwork  # noqa: F401
from .simple_neural_network import SimpleNeuralNetwork  # noqa: F401
