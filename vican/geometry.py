from vican_amd.geometry import (SE3, angle, deg2rad, distance_SO3, geodesic,  # noqa: F401
                                optimize_gauge_SE3, optimize_gauge_SO3, project_SO3, rad2deg)
