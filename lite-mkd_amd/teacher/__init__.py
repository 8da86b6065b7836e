from .mfm import ThreeTRXShiftLoopTime, ThreeTransforTemproal, TwoTransforFusion, TrainablePositionalEncoding  # noqa: F401
