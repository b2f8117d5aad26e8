"""String keys and defaults shared with the reference (bfcnn/constants.py:7-81); values must match."""

DEFAULT_EPSILON = 1e-3
DEFAULT_RELU_BIAS = 0.1
DEFAULT_BN_EPSILON = 1e-3
DEFAULT_LN_EPSILON = 1e-3
DEFAULT_BN_MOMENTUM = 0.995

TYPE_STR = "type"
MODEL_STR = "model"
CONFIG_STR = "config"
DATASET_STR = "dataset"
PARAMETERS_STR = "parameters"
BATCH_SIZE_STR = "batch_size"
INPUT_SHAPE_STR = "input_shape"
INPUT_TENSOR_STR = "input_tensor"
MODEL_HYDRA_DEFAULT_NAME_STR = "model_hydra.keras"

MAE_LOSS_STR = "mae_loss"
MSE_LOSS_STR = "mse_loss"
SSIM_LOSS_STR = "ssim_loss"
TOTAL_LOSS_STR = "total_loss"
REGULARIZATION_LOSS_STR = "regularization_loss"

USE_BIAS = "use_bias"
KERNEL_INITIALIZER = "kernel_initializer"
KERNEL_REGULARIZER = "kernel_regularizer"

BACKBONE_STR = "backbone"
DENOISER_STR = "denoiser"

MODEL_LOSS_FN_STR = "model"
DENOISER_LOSS_FN_STR = "denoiser"

CONFIG_PATH_STR = "config.json"

# files of this package's own model directory format (see model.save_model / load_model)
PIPELINE_FILE_STR = "pipeline.json"
WEIGHTS_FILE_STR = "weights.npz"
