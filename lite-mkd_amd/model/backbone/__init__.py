from .resnet import ResNet18Trunk  # noqa: F401
from .resnet18_2fc import resnet18_2fc  # noqa: F401
from .resnet18_student import resnet18_student  # noqa: F401
