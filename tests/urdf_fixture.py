"""The hand-written URDF fixture (tests/fixtures/lbr_iiwa14_like.urdf) registered as a buildable robot."""
import os

FIXTURE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fixtures", "lbr_iiwa14_like.urdf")
NAME = "iiwa14_urdf"


def register():
    from gridcodegenerator_amd import robots
    if NAME not in robots.REGISTERED_ROBOTS:
        robots.register_urdf(NAME, FIXTURE)
    return NAME
