from .gradient_optimizer import GradientOptimizer
from .basincma_optimizer import BasinCMAOptimizer
from .cma_optimizer import CMAOptimizer
from .ng_optimizer import NevergradOptimizer
from .hybrid_ng_optimizer import HybridNevergradOptimizer
