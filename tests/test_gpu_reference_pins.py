"""The reference's known-answer tests (tests/golden/reference_known_answers.json) run through
gridpp_amd -> C-ABI -> HIP kernels on a real MI355X."""
import pytest

from tests import pins

pytestmark = pytest.mark.gpu

# pins whose API is implemented on the GPU path so far
IMPLEMENTED = [
    "pin_barnes_basic", "pin_barnes_hmax", "pin_barnes_invalid",
    "pin_oi_simple_1d", "pin_oi_variance", "pin_oi_invalid_arguments", "pin_oi_missing_values",
    "pin_oi_extrapolation", "pin_oi_no_obs", "pin_radius_queries", "pin_invalid_coords", "pin_nearest",
    "pin_neighbourhood", "pin_neighbourhood_invalid", "pin_neighbourhood_3d_and_overflow", "pin_neighbourhood_quantile",
    "pin_neighbourhood_quantile_fast", "pin_thresholds", "pin_util", "pin_ensi", "pin_structures", "pin_oi_cross_validation",
    "pin_nearest_overloads", "pin_containers_next", "pin_gridding", "pin_count", "pin_distance", "pin_fill", "pin_doping", "pin_neighbourhood_search", "pin_calc_gradient", "pin_bilinear", "pin_bilinear_shapes", "pin_bilinear_missing_and_grids", "pin_point_in_rectangle", "pin_grid_get_box",
]


@pytest.fixture(scope="module")
def gridpp():
    import gridpp_amd
    assert gridpp_amd.device_count() >= 1, "no MI355X visible"
    return gridpp_amd


@pytest.mark.parametrize("name", IMPLEMENTED)
def test_gpu_pin(name, gridpp, golden):
    getattr(pins, name)(gridpp, golden)
