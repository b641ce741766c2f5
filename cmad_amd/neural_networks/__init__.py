from .input_convex_neural_network import AffineScaler, InputConvexNeuralNetwork  # noqa: F401
