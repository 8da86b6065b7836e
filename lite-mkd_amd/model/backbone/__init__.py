from .resnet import ResNet18Trunk, ResNet50Trunk  # noqa: F401
from .resnet50 import resnet50_2fc, resnet50_stduent  # noqa: F401
from .resnet18_2fc import resnet18_2fc  # noqa: F401
from .resnet18_student import resnet18_student  # noqa: F401
